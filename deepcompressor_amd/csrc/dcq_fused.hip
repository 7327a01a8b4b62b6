// Lean fused fake-quantization kernel for the common case (gfx950): integer quantization dtype,
// one scale level stored as float32 / float16 / bfloat16, optional zero point, groups of S elements
// contiguous in memory.  One read and one write of the tensor; HBM-bound by construction:
//
//   * a team of TL lanes (power of two) owns a group, every lane moves 16 bytes per access, so each
//     wave-instruction is a fully coalesced 1 KiB load or store;
//   * everything that depends on the configuration (storage type, zero-point domain, scale dtype) is a
//     template parameter: the per-element work is ~16 VALU instructions, well below the HBM time;
//   * min/max over the team by xor butterflies, scale / zero computed redundantly in registers.
//
// Arithmetic is the reference's, operation by operation (IEEE division, round-half-even, no FMA contraction).
#include "dcq_common.cuh"

namespace dcq {

struct FusedParams {
  const void* x;
  void* dequant;
  float* codes;
  float* scale_out;
  float* zero_out;
  const float* range_max;
  const float* range_min;
  int64_t G;
  int32_t S;
  int32_t team_lanes;
  int32_t team_shift;  // log2(team_lanes)
  int32_t chunks;
  int32_t has_ratio;
  float ratio;
  float q_lo, q_hi;     // tensor quant range
  float z_lo, z_hi;     // zero range (integer zero points)
  float s_lo, s_hi;     // scale range (integer scales)
  float q_span, q_min;
  int32_t has_bound_min, has_bound_max;
  float bound_min, bound_max;
  // last step of a progressive quantizer: the step sees clamp(x / outer[g], outer_lo, outer_hi) and its de-quantized
  // values are (optionally clamped, then) multiplied by outer[g] again
  const float* outer;
  float outer_lo, outer_hi;
  int32_t deq_clamp;
  float deq_lo, deq_hi;
  // |= kNonfiniteSeen when an element of x is NaN or +-Inf (NULL: not wanted)
  int32_t* nonfinite;
  // streaming (non-temporal) loads and stores: for tensors that do not fit the on-die caches anyway
  int32_t stream_hint;
};

// Tensors whose reads plus writes exceed this many bytes are moved with non-temporal loads and stores (measured on a
// 1.07 GB bf16 tensor, uint4 groups of 128: 186 -> 171 us; below the 256 MiB Infinity Cache a repeated pass is served
// on-die and the hint would throw that away).
constexpr int64_t kStreamBytes = int64_t{192} << 20;
constexpr int32_t kNonfiniteSeen = 4;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_piece(const uint4* p, bool nt) {
  if (nt) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
__device__ __forceinline__ float4 load_piece(const float4* p, bool nt) {
  if (nt) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
__device__ __forceinline__ void store_piece(uint4* p, const uint4& r, bool nt) {
  if (nt) {
    const u32x4_t v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
  } else {
    *p = r;
  }
}
__device__ __forceinline__ void store_piece(float4* p, const float4& r, bool nt) {
  if (nt) {
    const f32x4_t v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(p));
  } else {
    *p = r;
  }
}

// NaN / Inf detection that rides on the read: acc += x * 0 over every element is NaN as soon as one x is NaN or
// +-Inf (Inf * 0 = NaN) and stays 0 otherwise.  For the 16-bit types the two-element dot product instructions take
// the packed words as they are loaded: one instruction per two elements.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void report_nonfinite(float acc, int32_t* flags) {
  if (flags != nullptr && acc != acc) atomicOr(flags, kNonfiniteSeen);
}

__device__ __forceinline__ uint16_t bf16_rne_hw(float x) {
  const __bf16 h = static_cast<__bf16>(x);
  return __builtin_bit_cast(uint16_t, h);
}

template <int ST>
struct Elem;
template <>
struct Elem<DCQ_F32> {
  using raw_t = float4;  // 4 elements per 16 bytes
  static constexpr int kVec = 4;
  static __device__ __forceinline__ void unpack(const raw_t& r, float* v) {
    v[0] = r.x, v[1] = r.y, v[2] = r.z, v[3] = r.w;
  }
  static __device__ __forceinline__ raw_t pack(const float* v) { return make_float4(v[0], v[1], v[2], v[3]); }
  static __device__ __forceinline__ float probe(const raw_t& r, float acc) {
    acc = __builtin_fmaf(r.x, 0.0f, acc);
    acc = __builtin_fmaf(r.y, 0.0f, acc);
    acc = __builtin_fmaf(r.z, 0.0f, acc);
    return __builtin_fmaf(r.w, 0.0f, acc);
  }
};
template <>
struct Elem<DCQ_BF16> {
  using raw_t = uint4;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ void unpack(const raw_t& r, float* v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ raw_t pack(const float* v) {
    // two values per conversion instruction (v_cvt_pk_bf16_f32: round to nearest even, NaN kept), low half first
    typedef float pair_f32 __attribute__((ext_vector_type(2)));
    typedef __bf16 pair_bf16 __attribute__((ext_vector_type(2)));
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pair_f32{v[2 * i], v[2 * i + 1]}, pair_bf16));
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ float probe(const raw_t& r, float acc) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    const bf16x2_t zero = __builtin_bit_cast(bf16x2_t, 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[i]), zero, acc, false);
    return acc;
  }
};
template <>
struct Elem<DCQ_F16> {
  using raw_t = uint4;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ void unpack(const raw_t& r, float* v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
      const float2 f = __half22float2(h);
      v[2 * i] = f.x;
      v[2 * i + 1] = f.y;
    }
  }
  static __device__ __forceinline__ raw_t pack(const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __half2 h = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
      w[i] = *reinterpret_cast<const uint32_t*>(&h);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ float probe(const raw_t& r, float acc) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    const f16x2_t zero = __builtin_bit_cast(f16x2_t, 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w[i]), zero, acc, false);
    return acc;
  }
};

// xor-butterfly max / min over a team of TL adjacent lanes.  The first four steps stay inside a 16-lane row
// and are DPP operand modifiers (no LDS round trip): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then -- every
// quad being uniform by now -- row_half_mirror pairs the two quads of a half row and row_mirror the two halves.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// One butterfly step as a single VALU instruction: max / min of a lane's value with the DPP-permuted one.  (Written
// as asm because the compiler emits a v_mov_dpp plus two canonicalising v_max per step; the leading s_nop covers the
// two wait states a DPP read needs after the VALU write of its source, which the compiler cannot see inside asm.)
#define DCQ_DPP_STEP(NAME, CTRL)                                                                   \
  template <bool IS_MAX>                                                                           \
  __device__ __forceinline__ float NAME(float v) {                                                 \
    float r;                                                                                       \
    if (IS_MAX) {                                                                                  \
      asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
    } else {                                                                                       \
      asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
    }                                                                                              \
    return r;                                                                                      \
  }
DCQ_DPP_STEP(step_pairs, "quad_perm:[1,0,3,2]")
DCQ_DPP_STEP(step_quads, "quad_perm:[2,3,0,1]")
DCQ_DPP_STEP(step_half_rows, "row_half_mirror")
DCQ_DPP_STEP(step_rows, "row_mirror")
#undef DCQ_DPP_STEP

template <bool IS_MAX>
__device__ __forceinline__ float team_reduce(float v, int TL) {
  auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); };
  if (TL >= 2) v = step_pairs<IS_MAX>(v);
  if (TL >= 4) v = step_quads<IS_MAX>(v);
  if (TL >= 8) v = step_half_rows<IS_MAX>(v);
  if (TL >= 16) v = step_rows<IS_MAX>(v);
  if (TL >= 32) v = op(v, __shfl_xor(v, 16, kWave));
  if (TL >= 64) v = op(v, __shfl_xor(v, 32, kWave));
  return v;
}

__device__ __forceinline__ float clamp_f(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

// One rounding of a scale (or of a zero point kept in the scale's dtype); lo / hi only matter for an integer dtype.
template <int SK>
__device__ __forceinline__ float round_scale(float v, float lo, float hi) {
  if (SK == DCQ_ROUND_INT) return fminf(fmaxf(rintf(v), lo), hi);
  if (SK == DCQ_ROUND_F16) return f16_to_f32(f32_to_f16(v));
  if (SK == DCQ_ROUND_BF16) return __uint_as_float(static_cast<uint32_t>(bf16_rne_hw(v)) << 16);
  return v;
}

// groups a team of lanes holds at once in the one-chunk case; measured on MI355X (uint4 + zero point, groups of 128,
// bf16): 1 -> 5.5-5.6 TB/s, 2 -> 5.2, 4 -> 4.5 on 14336 x 4096
#ifndef DCQ_FUSED_U
#define DCQ_FUSED_U 1
#endif

// x[i] / outer for the elements of one 16-byte piece, clamped to the previous step's range
template <int VEC>
__device__ __forceinline__ void develop(float* v, float outer, const Divisor& d, const FusedParams& p) {
  float raw_abs = 0.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) raw_abs = fmaxf(raw_abs, fabsf(v[i]));
  if (d.fast && raw_abs <= 0x1p60f) {
#pragma unroll
    for (int i = 0; i < VEC; i += 2) {
      const f32x2 t = div_fast2(f32x2{v[i], v[i + 1]}, d);
      v[i] = t.x;
      v[i + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = v[i] / outer;
  }
  // med3 == fminf(fmaxf(v, lo), hi), NaN included (both give lo)
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = clamp_f(v[i], p.outer_lo, p.outer_hi);
}

// NP (one-chunk groups only): 16-byte pieces of its group a lane holds, pieces tl, tl + TL, ...; the team is then
// 1 / NP of the group's pieces wide, and the per-group work every lane repeats (reductions, scale, zero point,
// reciprocal) is issued once per NP pieces.
template <int ST, int ZD, int SK, bool SINGLE, bool OUTER, int NP>
__global__ __launch_bounds__(256) void fused_int_kernel(const FusedParams p) {
  using E = Elem<ST>;
  using raw_t = typename E::raw_t;
  constexpr int VEC = E::kVec;
  static_assert(SINGLE || NP == 1, "several pieces per lane only for groups a team holds whole");
  const int TL = p.team_lanes;
  const int tl = threadIdx.x & (TL - 1);
  const int64_t team0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> p.team_shift;
  const int64_t nteams = (static_cast<int64_t>(gridDim.x) * 256) >> p.team_shift;
  const raw_t* xin = static_cast<const raw_t*>(p.x);
  raw_t* dq_out = static_cast<raw_t*>(p.dequant);
  const int64_t raws_per_group = p.S / VEC;
  const bool nt = p.stream_hint != 0;
  float probe = 0.0f;

  constexpr int U = SINGLE ? DCQ_FUSED_U : 1;  // groups in flight per team: more bytes outstanding per lane
  for (int64_t g0 = team0; g0 < p.G; g0 += nteams * U) {
    raw_t keep[U][NP];
    float held[U][NP][VEC];  // one-chunk groups: the lane's values as the step sees them, computed once
    float s[U], z[U];
    Divisor dv[U];
    float outer[U];
    Divisor d_outer[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = g0 + u * nteams;
      live[u] = g < p.G;
      if (SINGLE && live[u]) {
#pragma unroll
        for (int h = 0; h < NP; ++h) {
          keep[u][h] = load_piece(xin + g * raws_per_group + h * TL + tl, nt);
          probe = E::probe(keep[u][h], probe);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = live[u] ? g0 + u * nteams : g0;
      const raw_t* xg = xin + g * raws_per_group;
      float vmax, vmin = 0.0f;
      float local_abs = INFINITY;  // bound on |x| over the elements THIS lane will quantize
      if (OUTER) {
        outer[u] = p.outer[g];
        d_outer[u] = make_divisor(outer[u], 0.0f);
      }
      if (p.range_max == nullptr) {
        float mx = ZD == DCQ_ZERO_NONE ? 0.0f : -INFINITY, mn = INFINITY;
        for (int c = 0; c < (SINGLE ? NP : p.chunks); ++c) {
          float v[VEC];
          if (SINGLE) {
            E::unpack(live[u] ? keep[u][c] : keep[0][c], v);
          } else {
            E::unpack(xg[static_cast<int64_t>(c) * TL + tl], v);
          }
          if (OUTER) develop<VEC>(v, outer[u], d_outer[u], p);
          if (SINGLE) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) held[u][c][i] = v[i];
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            if (ZD == DCQ_ZERO_NONE) {
              mx = fmaxf(mx, fabsf(v[i]));
            } else {
              mx = fmaxf(mx, v[i]);
              mn = fminf(mn, v[i]);
            }
          }
        }
        local_abs = ZD == DCQ_ZERO_NONE ? mx : fmaxf(fabsf(mx), fabsf(mn));
        vmax = team_reduce<true>(mx, TL);
        if (ZD != DCQ_ZERO_NONE) vmin = team_reduce<false>(mn, TL);
        if (p.has_ratio) {
          vmax = vmax * p.ratio;
          if (ZD != DCQ_ZERO_NONE) vmin = vmin * p.ratio;
        }
        if (ZD == DCQ_ZERO_PRE_SCALE) {
          vmax = fmaxf(vmax, 0.0f);
          vmin = fminf(vmin, 0.0f);
        }
      } else {
        vmax = p.range_max[g];
        if (ZD != DCQ_ZERO_NONE) vmin = p.range_min[g];
        if (SINGLE) {
          local_abs = 0.0f;
#pragma unroll
          for (int h = 0; h < NP; ++h) {
            float v[VEC];
            E::unpack(live[u] ? keep[u][h] : keep[0][h], v);
            if (OUTER) develop<VEC>(v, outer[u], d_outer[u], p);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              held[u][h][i] = v[i];
              local_abs = fmaxf(local_abs, fabsf(v[i]));
            }
          }
        }
      }
      if (p.has_bound_max) vmax = fminf(vmax, p.bound_max);
      if (ZD != DCQ_ZERO_NONE && p.has_bound_min) vmin = fmaxf(vmin, p.bound_min);
      const float span = ZD != DCQ_ZERO_NONE ? vmax - vmin : vmax;
      float sc = round_scale<SK>(fabsf(span / p.q_span), p.s_lo, p.s_hi);
      if (sc == 0.0f) sc = 1.0f;
      float ze = 0.0f;
      if (ZD == DCQ_ZERO_PRE_SCALE) ze = fminf(fmaxf(rintf(p.q_min - vmin / sc), p.z_lo), p.z_hi);
      if (ZD == DCQ_ZERO_POST_SCALE) ze = round_scale<SK>(p.q_min * sc - vmin, p.z_lo, p.z_hi);
      s[u] = sc;
      z[u] = ze;
      dv[u] = make_divisor(sc, local_abs + fabsf(ze));
      if (tl == 0 && live[u]) {
        if (p.scale_out) p.scale_out[g] = sc;
        if (p.zero_out) p.zero_out[g] = ze;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const int64_t g = g0 + u * nteams;
      const raw_t* xg = xin + g * raws_per_group;
      for (int c = 0; c < (SINGLE ? NP : p.chunks); ++c) {
        const int64_t off = static_cast<int64_t>(c) * TL + tl;
        float v[VEC], d[VEC], qv[VEC];
        if (SINGLE) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i] = held[u][c][i];
        } else {
          const raw_t piece = load_piece(xg + off, nt);
          probe = E::probe(piece, probe);
          E::unpack(piece, v);
          if (OUTER) develop<VEC>(v, outer[u], d_outer[u], p);
        }
        if (dv[u].fast) {
#pragma unroll
          for (int i = 0; i < VEC; i += 2) {
            f32x2 a = {v[i], v[i + 1]};
            const f32x2 zz = {z[u], z[u]}, ss = {s[u], s[u]};
            if (ZD == DCQ_ZERO_POST_SCALE) a = a + zz;
            f32x2 t = short_significand(SK) ? div_short2(a, dv[u]) : div_fast2(a, dv[u]);
            if (ZD == DCQ_ZERO_PRE_SCALE) t = t + zz;
            f32x2 q = {clamp_f(rintf(t.x), p.q_lo, p.q_hi), clamp_f(rintf(t.y), p.q_lo, p.q_hi)};
            qv[i] = q.x;
            qv[i + 1] = q.y;
            f32x2 o;
            if (ZD == DCQ_ZERO_PRE_SCALE) {
              o = (q - zz) * ss;
            } else if (ZD == DCQ_ZERO_POST_SCALE) {
              o = q * ss - zz;
            } else {
              o = q * ss;
            }
            d[i] = o.x;
            d[i + 1] = o.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float t = to_code_input<ZD>(v[i], s[u], z[u]);
            qv[i] = fminf(fmaxf(rintf(t), p.q_lo), p.q_hi);
            d[i] = from_code<ZD>(qv[i], s[u], z[u]);
          }
        }
        if (OUTER) {
          if (p.deq_clamp) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) d[i] = clamp_f(d[i], p.deq_lo, p.deq_hi);
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) d[i] = d[i] * outer[u];
        }
        if (dq_out) store_piece(dq_out + g * raws_per_group + off, E::pack(d), nt);
        if (p.codes) {
          float4* co = reinterpret_cast<float4*>(p.codes + (g * raws_per_group + off) * VEC);
#pragma unroll
          for (int i = 0; i < VEC / 4; ++i)
            store_piece(co + i, make_float4(qv[4 * i], qv[4 * i + 1], qv[4 * i + 2], qv[4 * i + 3]), nt);
        }
      }
    }
  }
  report_nonfinite(probe, p.nonfinite);
}

template <int ST, int ZD, int SK>
static int launch_fused(const FusedParams& pin, hipStream_t stream) {
  FusedParams p = pin;
  // one-chunk groups of several pieces: a lane takes up to 4 pieces of its group (see the kernel's NP)
  // (measured on a 32768 x 8192 bf16 tensor, uint4 + zero point, groups of 128: 1 piece per lane 232 us, 2: 219 us,
  // 4: 219 us)
  int np = 1;
  if (p.chunks == 1 && p.team_lanes >= 2) {
    np = 2;
    p.team_lanes /= 2;
    p.team_shift -= 1;
  }
  const int64_t threads = p.G * p.team_lanes;
  int64_t blocks = (threads + 255) / 256;
  if (p.chunks == 1) blocks = (blocks + DCQ_FUSED_U - 1) / DCQ_FUSED_U;  // several groups per team and pass
  // measured on MI355X (32768 x 8192 bf16, uint4 groups of 128, a lane holding two pieces): a team per group with no
  // loop streams fastest -- 202 us on a persistent grid of 8192 workgroups, 190 / 186 us on 32768 / 65536 (one group
  // per team), and with non-temporal loads and stores 190 / 176 / 171 us against 164 us for a plain copy of the same
  // bytes.  Groups longer than a team (several chunks) keep the persistent grid.
  const int64_t kGridCap = p.chunks == 1 ? (int64_t{1} << 22) : 8192;
  if (blocks > kGridCap) blocks = kGridCap;
  const int64_t moved = p.G * p.S * ((p.codes ? 4 : 0) + (p.dequant ? (ST == DCQ_F32 ? 4 : 2) : 0) + (ST == DCQ_F32 ? 4 : 2));
  p.stream_hint = moved >= kStreamBytes;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define DCQ_LAUNCH_SINGLE(OUTER_)                                                                        \
  do {                                                                                                   \
    if (np == 2) {                                                                                       \
      hipLaunchKernelGGL((fused_int_kernel<ST, ZD, SK, true, OUTER_, 2>), grid, block, 0, stream, p);    \
    } else {                                                                                             \
      hipLaunchKernelGGL((fused_int_kernel<ST, ZD, SK, true, OUTER_, 1>), grid, block, 0, stream, p);    \
    }                                                                                                    \
  } while (0)
  if (p.outer) {
    if (p.chunks == 1) {
      DCQ_LAUNCH_SINGLE(true);
    } else {
      hipLaunchKernelGGL((fused_int_kernel<ST, ZD, SK, false, true, 1>), grid, block, 0, stream, p);
    }
  } else if (p.chunks == 1) {
    DCQ_LAUNCH_SINGLE(false);
  } else {
    hipLaunchKernelGGL((fused_int_kernel<ST, ZD, SK, false, false, 1>), grid, block, 0, stream, p);
  }
#undef DCQ_LAUNCH_SINGLE
  DCQ_LAUNCH_CHECK("fused_int_kernel launch");
  return DCQ_OK;
}

template <int ST, int ZD>
static int fused_scale_kind(const FusedParams& p, int sk, hipStream_t stream) {
  switch (sk) {
    case DCQ_ROUND_F16:
      return launch_fused<ST, ZD, DCQ_ROUND_F16>(p, stream);
    case DCQ_ROUND_BF16:
      return launch_fused<ST, ZD, DCQ_ROUND_BF16>(p, stream);
    case DCQ_ROUND_INT:
      return launch_fused<ST, ZD, DCQ_ROUND_INT>(p, stream);
    default:
      return launch_fused<ST, ZD, DCQ_ROUND_F32>(p, stream);
  }
}

template <int ST>
static int fused_zero(const FusedParams& p, int zd, int sk, hipStream_t stream) {
  switch (zd) {
    case DCQ_ZERO_NONE:
      return fused_scale_kind<ST, DCQ_ZERO_NONE>(p, sk, stream);
    case DCQ_ZERO_PRE_SCALE:
      return fused_scale_kind<ST, DCQ_ZERO_PRE_SCALE>(p, sk, stream);
    default:
      return fused_scale_kind<ST, DCQ_ZERO_POST_SCALE>(p, sk, stream);
  }
}

// Returns DCQ_ERR_UNSUPPORTED (without setting an error) when the configuration or the buffers do not fit the
// lean kernel; the caller then uses the general kernel.
int try_launch_fused_int(const dcq_step& st, const void* x, int storage, int64_t G, int64_t S, const float* range_max,
                         const float* range_min, int has_ratio, float ratio, void* dequant, float* codes, float* scale,
                         float* zero, hipStream_t stream, const OuterStep* outer, int32_t* nonfinite) {
  const int sk = st.scale.kind;
  const bool scale_ok =
      (sk == DCQ_ROUND_F32 || sk == DCQ_ROUND_F16 || sk == DCQ_ROUND_BF16 || sk == DCQ_ROUND_INT) && !st.scale.has_clamp;
  bool zero_ok = true;
  if (st.zero_domain == DCQ_ZERO_PRE_SCALE) zero_ok = st.zero.kind == DCQ_ROUND_INT;
  if (st.zero_domain == DCQ_ZERO_POST_SCALE) zero_ok = st.zero.kind == sk && !st.zero.has_clamp;
  if (st.tensor.kind != DCQ_ROUND_INT || !scale_ok || !zero_ok) return DCQ_ERR_UNSUPPORTED;
  const int vec = storage == DCQ_F32 ? 4 : 8;
  auto aligned = [](const void* ptr, uintptr_t a) { return ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) % a) == 0; };
  if (S % vec != 0 || !aligned(x, 16) || !aligned(dequant, 16) || !aligned(codes, 16) || S > (1 << 30))
    return DCQ_ERR_UNSUPPORTED;
  const int64_t n = S / vec;
  int lanes = 1;
  while (lanes < 64 && n % (lanes * 2) == 0) lanes *= 2;
  FusedParams p{};
  p.x = x;
  p.dequant = dequant;
  p.codes = codes;
  p.scale_out = scale;
  p.zero_out = zero;
  p.range_max = range_max;
  p.range_min = range_min;
  p.G = G;
  p.S = static_cast<int32_t>(S);
  p.team_lanes = lanes;
  p.team_shift = __builtin_ctz(static_cast<unsigned>(lanes));
  p.chunks = static_cast<int32_t>(n / lanes);
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.q_lo = st.tensor.lo;
  p.q_hi = st.tensor.hi;
  p.z_lo = st.zero.lo;
  p.z_hi = st.zero.hi;
  p.s_lo = st.scale.lo;
  p.s_hi = st.scale.hi;
  p.q_span = st.q_span;
  p.q_min = st.q_min;
  p.has_bound_min = st.has_bound_min;
  p.has_bound_max = st.has_bound_max;
  p.bound_min = st.bound_min;
  p.bound_max = st.bound_max;
  p.nonfinite = nonfinite;
  if (outer != nullptr && outer->scale != nullptr) {
    p.outer = outer->scale;
    p.outer_lo = outer->lo;
    p.outer_hi = outer->hi;
    p.deq_clamp = outer->deq_clamp;
    p.deq_lo = outer->deq_lo;
    p.deq_hi = outer->deq_hi;
  }
  switch (storage) {
    case DCQ_F32:
      return fused_zero<DCQ_F32>(p, st.zero_domain, sk, stream);
    case DCQ_F16:
      return fused_zero<DCQ_F16>(p, st.zero_domain, sk, stream);
    default:
      return fused_zero<DCQ_BF16>(p, st.zero_domain, sk, stream);
  }
}

// =============================================================================================
// Two progressive steps in one pass (QoQ-style W4: a float scale per row from the row's largest magnitude, then an
// integer step on groups of clamp(x / row scale) inside the row): one workgroup holds a row in registers, reduces
// its largest magnitude through LDS, and every team of lanes then runs the last step on its group exactly as
// fused_int_kernel<OUTER> does -- the row scale never leaves the chip and the tensor is read once.
// =============================================================================================
constexpr int kMaxRowPieces = 8;  // 16-byte pieces per thread: rows of up to 256 * 8 * (4 | 8) values

struct ProgressiveParams {
  FusedParams last;        // last.outer unused (the row scale is computed here); last.outer_lo / hi / deq_* used
  dcq_rounding first_scale;
  float first_q_span;
  int32_t first_has_ratio;
  float first_ratio;
  int32_t first_has_bound_max;
  float first_bound_max;
  float* first_scale_out;  // [rows]
  int64_t rows;
  int32_t row_pieces;      // 16-byte pieces per row
  int32_t groups_per_row;
};

// NP: 16-byte pieces of ONE group a lane holds.  With NP = 2 a group's pieces are shared by half as many lanes (a
// team of TL / 2, each with pieces i and i + TL / 2 of the group), so the per-group work every lane repeats -- the
// reductions, scale, zero point, reciprocal -- is issued once per 16 or 32 values instead of once per 8 or 16; a
// wave-instruction still reads whole 128-byte lines (8 lanes x 16 bytes for groups of 128 values).
template <int ST, int ZD, int SK, int PIECES, int NP>
__global__ __launch_bounds__(256) void progressive_int_kernel(const ProgressiveParams pp) {
  using E = Elem<ST>;
  using raw_t = typename E::raw_t;
  constexpr int VEC = E::kVec;
  const FusedParams& p = pp.last;
  __shared__ float part[2][4];
  const int TL = p.team_lanes / NP;           // lanes per team
  const int team_shift = p.team_shift - (NP == 4 ? 2 : NP - 1);
  const int tid = threadIdx.x;
  const raw_t* xin = static_cast<const raw_t*>(p.x);
  raw_t* dq_out = static_cast<raw_t*>(p.dequant);
  const Divisor d_span = make_divisor(p.q_span, 0.0f);
  // piece held in slot c: slots c .. c + NP - 1 (c a multiple of NP) belong to group (c / NP) * (256 / TL) + tid / TL
  // of the row, whose pieces h * TL + tid % TL, h < NP, they are
  auto piece_of = [&](int c) __attribute__((always_inline)) {
    if (NP == 1) return c * 256 + tid;
    const int group = (c / NP) * (256 >> team_shift) + (tid >> team_shift);
    return (group << p.team_shift) + (c % NP) * TL + (tid & (TL - 1));
  };
  int parity = 0;
  const bool nt = p.stream_hint != 0;
  float probe = 0.0f;
  for (int64_t row = blockIdx.x; row < pp.rows; row += gridDim.x, parity ^= 1) {
    const int64_t base = row * pp.row_pieces;
    raw_t keep[PIECES];
    float mx = 0.0f;
#pragma unroll
    for (int c = 0; c < PIECES; ++c) {
      const int j = piece_of(c);
      if (j < pp.row_pieces) {
        keep[c] = load_piece(xin + base + j, nt);
        probe = E::probe(keep[c], probe);
        float v[VEC];
        E::unpack(keep[c], v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) mx = fmaxf(mx, fabsf(v[i]));
      }
    }
    mx = team_reduce<true>(mx, 64);
    if ((tid & 63) == 0) part[parity][tid >> 6] = mx;
    __syncthreads();  // one barrier per row: the next row writes the other half of `part`
    float top = fmaxf(fmaxf(part[parity][0], part[parity][1]), fmaxf(part[parity][2], part[parity][3]));
    const float row_abs = top;  // largest magnitude of the row as stored
    if (pp.first_has_ratio) top = top * pp.first_ratio;
    if (pp.first_has_bound_max) top = fminf(top, pp.first_bound_max);
    float outer = apply_rounding(pp.first_scale, fabsf(top / pp.first_q_span), nullptr);
    if (outer == 0.0f) outer = 1.0f;
    // the same value in every lane (it came through LDS): tell the compiler, so that what depends on it alone is
    // scalar work and scalar branches
    outer = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(outer)));
    if (tid == 0) pp.first_scale_out[row] = outer;
    const Divisor d_outer = make_divisor(outer, 0.0f);
    // the hoisted-reciprocal division's guard (see make_divisor) for the whole row at once
    const bool row_fast = __builtin_amdgcn_readfirstlane(static_cast<int>(d_outer.fast && row_abs <= 0x1p60f)) != 0;
    const bool short_outer = short_significand(pp.first_scale.kind);  // the row scale was rounded to that dtype
#pragma unroll
    for (int c = 0; c < PIECES; c += NP) {
      const int j0 = piece_of(c);
      if (j0 >= pp.row_pieces) continue;  // whole teams leave together: a group never straddles the row's end
      float v[NP][VEC], d[NP][VEC], qv[NP][VEC];
      float lmx = ZD == DCQ_ZERO_NONE ? 0.0f : -INFINITY, lmn = INFINITY;
#pragma unroll
      for (int h = 0; h < NP; ++h) {
        E::unpack(keep[c + h], v[h]);
        if (row_fast) {
#pragma unroll
          for (int i = 0; i < VEC; i += 2) {
            const f32x2 x2 = {v[h][i], v[h][i + 1]};
            const f32x2 t = short_outer ? div_short2(x2, d_outer) : div_fast2(x2, d_outer);
            v[h][i] = clamp_f(t.x, p.outer_lo, p.outer_hi);
            v[h][i + 1] = clamp_f(t.y, p.outer_lo, p.outer_hi);
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[h][i] = clamp_f(v[h][i] / outer, p.outer_lo, p.outer_hi);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (ZD == DCQ_ZERO_NONE) {
            lmx = fmaxf(lmx, fabsf(v[h][i]));
          } else {
            lmx = fmaxf(lmx, v[h][i]);
            lmn = fminf(lmn, v[h][i]);
          }
        }
      }
      const float local_abs = ZD == DCQ_ZERO_NONE ? lmx : fmaxf(fabsf(lmx), fabsf(lmn));
      float vmax = team_reduce<true>(lmx, TL), vmin = 0.0f;
      if (ZD != DCQ_ZERO_NONE) vmin = team_reduce<false>(lmn, TL);
      if (p.has_ratio) {
        vmax = vmax * p.ratio;
        if (ZD != DCQ_ZERO_NONE) vmin = vmin * p.ratio;
      }
      if (ZD == DCQ_ZERO_PRE_SCALE) {
        vmax = fmaxf(vmax, 0.0f);
        vmin = fminf(vmin, 0.0f);
      }
      if (p.has_bound_max) vmax = fminf(vmax, p.bound_max);
      if (ZD != DCQ_ZERO_NONE && p.has_bound_min) vmin = fmaxf(vmin, p.bound_min);
      const float span = ZD != DCQ_ZERO_NONE ? vmax - vmin : vmax;
      // span / q_span: the divisor is the same for every group, the guard is voted on by the wave
      float ratio_to_span;
      if (__builtin_amdgcn_ballot_w64(!(d_span.fast && fabsf(span) <= 0x1p60f)) == 0) {
        ratio_to_span = div_fast(span, d_span);
      } else {
        ratio_to_span = span / p.q_span;
      }
      float sc = round_scale<SK>(fabsf(ratio_to_span), p.s_lo, p.s_hi);
      if (sc == 0.0f) sc = 1.0f;
      float ze = 0.0f;
      if (ZD == DCQ_ZERO_PRE_SCALE) ze = fminf(fmaxf(rintf(p.q_min - vmin / sc), p.z_lo), p.z_hi);
      if (ZD == DCQ_ZERO_POST_SCALE) ze = round_scale<SK>(p.q_min * sc - vmin, p.z_lo, p.z_hi);
      const Divisor dv = make_divisor(sc, local_abs + fabsf(ze));
      const int64_t g = row * pp.groups_per_row + (j0 >> p.team_shift);
      if ((tid & (TL - 1)) == 0) {
        if (p.scale_out) p.scale_out[g] = sc;
        if (p.zero_out) p.zero_out[g] = ze;
      }
      const bool fast_codes = __builtin_amdgcn_ballot_w64(!dv.fast) == 0;
#pragma unroll
      for (int h = 0; h < NP; ++h) {
        if (fast_codes) {
#pragma unroll
          for (int i = 0; i < VEC; i += 2) {
            f32x2 a = {v[h][i], v[h][i + 1]};
            const f32x2 zz = {ze, ze}, ss = {sc, sc};
            if (ZD == DCQ_ZERO_POST_SCALE) a = a + zz;
            f32x2 t = short_significand(SK) ? div_short2(a, dv) : div_fast2(a, dv);
            if (ZD == DCQ_ZERO_PRE_SCALE) t = t + zz;
            const f32x2 q = {clamp_f(rintf(t.x), p.q_lo, p.q_hi), clamp_f(rintf(t.y), p.q_lo, p.q_hi)};
            qv[h][i] = q.x;
            qv[h][i + 1] = q.y;
            f32x2 o;
            if (ZD == DCQ_ZERO_PRE_SCALE) {
              o = (q - zz) * ss;
            } else if (ZD == DCQ_ZERO_POST_SCALE) {
              o = q * ss - zz;
            } else {
              o = q * ss;
            }
            d[h][i] = o.x;
            d[h][i + 1] = o.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float t = to_code_input<ZD>(v[h][i], sc, ze);
            qv[h][i] = fminf(fmaxf(rintf(t), p.q_lo), p.q_hi);
            d[h][i] = from_code<ZD>(qv[h][i], sc, ze);
          }
        }
        if (p.deq_clamp) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) d[h][i] = clamp_f(d[h][i], p.deq_lo, p.deq_hi);
        }
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
          const f32x2 o = f32x2{d[h][i], d[h][i + 1]} * f32x2{outer, outer};
          d[h][i] = o.x;
          d[h][i + 1] = o.y;
        }
        const int64_t at = base + j0 + h * TL;
        if (dq_out) store_piece(dq_out + at, E::pack(d[h]), nt);
        if (p.codes) {
          float4* co = reinterpret_cast<float4*>(p.codes + at * VEC);
#pragma unroll
          for (int i = 0; i < VEC / 4; ++i)
            store_piece(co + i, make_float4(qv[h][4 * i], qv[h][4 * i + 1], qv[h][4 * i + 2], qv[h][4 * i + 3]), nt);
        }
      }
    }
  }
  report_nonfinite(probe, p.nonfinite);
}

// PIECES: 16-byte pieces a thread holds (the row in registers); fewer registers, more waves in flight
template <int ST, int ZD, int SK, int NP>
static int progressive_launch_np(const ProgressiveParams& pp, hipStream_t stream) {
  // a workgroup per row, no loop up to 2^20 rows (measured on 32768 rows of 8192 bf16 values: 236 us on a persistent
  // grid of 8192 workgroups, 226 us with one workgroup per row).  The kernel is bound by its phases, not by the
  // vector pipe or HBM: a workgroup loads its row, waits for the row's largest magnitude, then computes and stores,
  // and at 95 registers per lane five workgroups share a compute unit (fewer pieces of a group per lane: 84 / 67
  // registers, but more per-group work: 244 / 258 us).
  const ProgressiveParams& pp2 = pp;
  const dim3 grid(static_cast<unsigned>(pp.rows < (1 << 20) ? pp.rows : (1 << 20))), block(256);
  if (pp.row_pieces <= 256 * 2) {
    hipLaunchKernelGGL((progressive_int_kernel<ST, ZD, SK, 2, (NP > 2 ? 2 : NP)>), grid, block, 0, stream, pp2);
  } else if (pp.row_pieces <= 256 * 4) {
    hipLaunchKernelGGL((progressive_int_kernel<ST, ZD, SK, 4, NP>), grid, block, 0, stream, pp2);
  } else {
    hipLaunchKernelGGL((progressive_int_kernel<ST, ZD, SK, kMaxRowPieces, NP>), grid, block, 0, stream, pp2);
  }
  DCQ_LAUNCH_CHECK("progressive_int_kernel launch");
  return DCQ_OK;
}

template <int ST, int ZD, int SK>
static int progressive_launch(const ProgressiveParams& pp, hipStream_t stream) {
  // a lane takes four (two) pieces of its group when the group has that many (measured on 32768 x 8192 bf16, QoQ W4
  // with groups of 128: 264 us with one piece per lane, 245 with two, 232 with four)
  if (pp.last.team_lanes >= 4) return progressive_launch_np<ST, ZD, SK, 4>(pp, stream);
  if (pp.last.team_lanes >= 2) return progressive_launch_np<ST, ZD, SK, 2>(pp, stream);
  return progressive_launch_np<ST, ZD, SK, 1>(pp, stream);
}

template <int ST, int ZD>
static int progressive_scale_kind(const ProgressiveParams& pp, int sk, hipStream_t stream) {
  switch (sk) {
    case DCQ_ROUND_F16:
      return progressive_launch<ST, ZD, DCQ_ROUND_F16>(pp, stream);
    case DCQ_ROUND_BF16:
      return progressive_launch<ST, ZD, DCQ_ROUND_BF16>(pp, stream);
    case DCQ_ROUND_INT:
      return progressive_launch<ST, ZD, DCQ_ROUND_INT>(pp, stream);
    default:
      return progressive_launch<ST, ZD, DCQ_ROUND_F32>(pp, stream);
  }
}

template <int ST>
static int progressive_zero(const ProgressiveParams& pp, int zd, int sk, hipStream_t stream) {
  switch (zd) {
    case DCQ_ZERO_NONE:
      return progressive_scale_kind<ST, DCQ_ZERO_NONE>(pp, sk, stream);
    case DCQ_ZERO_PRE_SCALE:
      return progressive_scale_kind<ST, DCQ_ZERO_PRE_SCALE>(pp, sk, stream);
    default:
      return progressive_scale_kind<ST, DCQ_ZERO_POST_SCALE>(pp, sk, stream);
  }
}

static bool lean_int_step(const dcq_step& st) {
  const int sk = st.scale.kind;
  const bool scale_ok =
      (sk == DCQ_ROUND_F32 || sk == DCQ_ROUND_F16 || sk == DCQ_ROUND_BF16 || sk == DCQ_ROUND_INT) && !st.scale.has_clamp;
  bool zero_ok = true;
  if (st.zero_domain == DCQ_ZERO_PRE_SCALE) zero_ok = st.zero.kind == DCQ_ROUND_INT;
  if (st.zero_domain == DCQ_ZERO_POST_SCALE) zero_ok = st.zero.kind == sk && !st.zero.has_clamp;
  return st.tensor.kind == DCQ_ROUND_INT && scale_ok && zero_ok;
}

}  // namespace dcq

extern "C" int dcq_fake_quant_progressive(const dcq_step* first, const dcq_step* last, const void* x, int32_t storage,
                                          int64_t rows, int64_t row_len, int64_t S, int32_t first_has_ratio,
                                          float first_ratio, int32_t has_ratio, float ratio, float outer_lo,
                                          float outer_hi, int32_t deq_clamp, float deq_lo, float deq_hi, void* dequant,
                                          float* codes, float* first_scale, float* scale, float* zero,
                                          int32_t* nonfinite, dcq_stream_t stream) {
  using namespace dcq;
  DCQ_REQUIRE(first && last, DCQ_ERR_ARGUMENT, "step is NULL");
  DCQ_REQUIRE(rows >= 0 && row_len >= 0 && S > 0, DCQ_ERR_ARGUMENT, "bad size");
  if (rows == 0 || row_len == 0) return DCQ_OK;
  DCQ_REQUIRE(x && first_scale, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(storage >= DCQ_F32 && storage <= DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage %d", storage);
  DCQ_REQUIRE(first->zero_domain == DCQ_ZERO_NONE && first->scale.kind != DCQ_ROUND_CODEBOOK, DCQ_ERR_UNSUPPORTED,
              "the first step must have no zero point and a scale that is not a value table");
  DCQ_REQUIRE(lean_int_step(*last), DCQ_ERR_UNSUPPORTED,
              "the last step must be an integer type with one float or integer scale level");
  const int vec = storage == DCQ_F32 ? 4 : 8;
  auto aligned = [](const void* ptr) { return ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) % 16) == 0; };
  DCQ_REQUIRE(aligned(x) && aligned(dequant) && aligned(codes), DCQ_ERR_UNSUPPORTED, "buffers must be 16-byte aligned");
  const int64_t lanes = S / vec;
  DCQ_REQUIRE(S % vec == 0 && row_len % S == 0 && lanes <= 64 && (lanes & (lanes - 1)) == 0, DCQ_ERR_UNSUPPORTED,
              "groups must be 1..64 whole 16-byte pieces (a power of two) and tile the row");
  DCQ_REQUIRE(row_len / vec <= 256 * kMaxRowPieces, DCQ_ERR_UNSUPPORTED, "rows of more than %d values",
              256 * kMaxRowPieces * vec);
  ProgressiveParams pp{};
  FusedParams& p = pp.last;
  p.x = x;
  p.dequant = dequant;
  p.codes = codes;
  p.scale_out = scale;
  p.zero_out = zero;
  p.G = rows * (row_len / S);
  p.S = static_cast<int32_t>(S);
  p.team_lanes = static_cast<int32_t>(lanes);
  p.team_shift = __builtin_ctz(static_cast<unsigned>(lanes));
  p.chunks = 1;
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.q_lo = last->tensor.lo;
  p.q_hi = last->tensor.hi;
  p.z_lo = last->zero.lo;
  p.z_hi = last->zero.hi;
  p.s_lo = last->scale.lo;
  p.s_hi = last->scale.hi;
  p.q_span = last->q_span;
  p.q_min = last->q_min;
  p.has_bound_min = last->has_bound_min;
  p.has_bound_max = last->has_bound_max;
  p.bound_min = last->bound_min;
  p.bound_max = last->bound_max;
  p.outer_lo = outer_lo;
  p.outer_hi = outer_hi;
  p.deq_clamp = deq_clamp;
  p.deq_lo = deq_lo;
  p.deq_hi = deq_hi;
  p.nonfinite = nonfinite;
  // no streaming hint here: with four pieces of a group per lane a wave-instruction moves 64-byte half lines, and the
  // other half must still be in the cache when the next instruction asks for it (measured with the hint: 226 -> 254 us)
  p.stream_hint = 0;
  pp.first_scale = first->scale;
  pp.first_q_span = first->q_span;
  pp.first_has_ratio = first_has_ratio;
  pp.first_ratio = first_ratio;
  pp.first_has_bound_max = first->has_bound_max;
  pp.first_bound_max = first->bound_max;
  pp.first_scale_out = first_scale;
  pp.rows = rows;
  pp.row_pieces = static_cast<int32_t>(row_len / vec);
  pp.groups_per_row = static_cast<int32_t>(row_len / S);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (storage) {
    case DCQ_F32:
      return progressive_zero<DCQ_F32>(pp, last->zero_domain, last->scale.kind, s);
    case DCQ_F16:
      return progressive_zero<DCQ_F16>(pp, last->zero_domain, last->scale.kind, s);
    default:
      return progressive_zero<DCQ_BF16>(pp, last->zero_domain, last->scale.kind, s);
  }
}

namespace dcq {

// =============================================================================================
// Floating-point quantization dtypes (value tables such as fp4 e2m1) without a zero point, the scale in one level or
// in two with the first level per tensor (NVFP4: the largest group scale of the tensor, float; every group's scale
// relative to it, fp8).  Same single read and single write as the kernel above.  Two things keep it off the VALU:
//
//   * rounding to the table is one LDS lookup, not a search: when every table value is a multiple of a power of two
//     `cell` (true of minifloat tables: 0.5 for e2m1), the two neighbours of any x in [k cell, (k+1) cell) are the
//     same pair, so a list of (below, above) per cell, built in LDS by each workgroup from the table, leaves one
//     comparison x + x < below + above per element -- the reference's own tie rule (ties go up).  A table that is not
//     on such a grid (or needs more than kMaxCells cells) is searched as in the general kernel;
//   * groups of two 16-byte pieces (16 values of a 16-bit tensor) are owned by ONE lane, so the scale -- two divisions
//     and a search of the fp8 table -- is computed once per group instead of once per piece.  Loads and stores stay
//     fully coalesced: a wave moves 128 consecutive pieces with two instructions and neighbouring lanes swap one piece
//     each (a DPP move) so that even lanes hold groups 0..31 and odd lanes groups 32..63 of the wave's 64.
// =============================================================================================
constexpr int kMaxValueCells = 2048;  // 8 KiB: half-cells of the value table (e2m1: 48)
constexpr int kMaxScaleCells = 1024;  // 4 KiB: cells of the scale table by exponent and leading mantissa bits (e4m3: 306)

struct CodebookParams {
  dcq_step step;
  const void* x;
  void* dequant;
  float* codes;
  float* scale_out;
  float* zero_out;  // receives 0 for every group (there is no zero point)
  int64_t G;
  int32_t S;
  int32_t team_lanes;
  int32_t team_shift;
  int32_t has_ratio;
  float ratio;
  // two scale levels: device scalar holding the tensor's largest magnitude, the first level's rounding and span
  const float* tensor_absmax;
  dcq_rounding level0;
  float level0_span;
  float* level0_out;
  float* level1_out;
  int32_t* nonfinite;
  int32_t stream_hint;
  int32_t passes;       // consecutive passes of a workgroup (its share of the tensor is one contiguous run)
};

struct CodebookLut {
  float tensor[256];
  float scale[256];
  float value[kMaxValueCells + 1];
  float scale_value[kMaxScaleCells + 2];
  // per wave: min over the non-zero table values of the exponent of their lowest set bit; min over the positive
  // scale-table values and midpoints of the trailing zero bits; the same over the values alone (24 minus this =
  // bits of a scale's significand); bit pattern of the smallest positive midpoint of the scale table
  unsigned reduce[4][4];
};

// How a value is rounded to the tensor table / a non-negative value to the scale table (0 cells: search the table)
struct GridRound {
  float inv_half;   // 1 / (half a cell of the value table)
  int first;        // index of the table's first value in half-cells
  int cells;
  unsigned scale_shift;
  int scale_base;   // key of the scale table's cell 0
  int scale_cells;
  int scale_bits;   // significand bits of a value of the scale's dtype (24: unknown / float32)
};

__device__ __forceinline__ float table_nearest(float x, const float* cb, int C) { return codebook_nearest(x, cb, C); }

// Stages the tables and builds the lookup lists.
//
// Value table: when every table value is a multiple of a power of two `cell` (true of minifloat tables: 0.5 for
// e2m1), every value AND every midpoint between neighbours is a multiple of cell / 2, so all x in
// [k cell / 2, (k + 1) cell / 2) round to the same value under the reference's rule (lo if x + x < lo + hi, else hi:
// the breakpoints are the midpoints, a midpoint itself goes up): one list entry per half-cell, filled by applying
// that rule to the half-cell's lower edge.  x / (cell / 2) is exact (a power of two) and so is its floor.
//
// Scale table (inputs are magnitudes): the same idea on the bit pattern.  With `shift` = the smallest number of
// trailing zero bits over the positive table values and positive midpoints, every breakpoint is the lower edge of a
// cell {x : bits(x) >> shift == k}; below the smallest positive midpoint the result is the table's zero.
// Tables that need more cells than the lists hold are searched.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = min(v, static_cast<unsigned>(__shfl_xor(static_cast<int>(v), off, kWave)));
  return v;
}

__device__ __forceinline__ GridRound build_lut(const CodebookParams& p, CodebookLut& lds) {
  const dcq_rounding& rt = p.step.tensor;
  const dcq_rounding& rs = p.step.scale;
  const int C = rt.codebook_size;
  const bool scale_table = rs.kind == DCQ_ROUND_CODEBOOK;
  const int CS = scale_table ? rs.codebook_size : 0;
  // one table entry per thread (tables hold at most 256 values, a workgroup has 256 threads); the four minima the
  // lists need are reduced inside each wave and meet in LDS: two barriers in all, no atomics
  const int t = threadIdx.x;
  unsigned low = 0xffffffffu;  // (exponent of the lowest set bit of a table value) + 1024
  unsigned ctz_all = 23, ctz_val = 23, first_mid = 0x7f800000u;
  if (t < C) {
    const float v = rt.codebook[t];
    lds.tensor[t] = v;
    const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
    if (b != 0 && b < 0x7f800000u) {
      const int e = static_cast<int>(b >> 23);
      const uint32_t m = (b & 0x7fffffu) | (e ? 0x800000u : 0u);
      low = static_cast<unsigned>((e ? e : 1) - 150 + __builtin_ctz(m) + 1024);
    }
  }
  if (t < CS) {
    const float v = rs.codebook[t];
    lds.scale[t] = v;
    if (v > 0.0f && v < INFINITY) {
      ctz_val = static_cast<unsigned>(__builtin_ctz(__float_as_uint(v)));
      ctz_all = ctz_val;
      if (t > 0) {
        const float mid = (rs.codebook[t - 1] + v) * 0.5f;
        if (mid > 0.0f) {
          ctz_all = min(ctz_all, static_cast<unsigned>(__builtin_ctz(__float_as_uint(mid))));
          first_mid = __float_as_uint(mid);  // positive floats order like their bit patterns
        }
      }
    }
  }
  low = wave_min_u32(low);
  ctz_all = wave_min_u32(ctz_all);
  ctz_val = wave_min_u32(ctz_val);
  first_mid = wave_min_u32(first_mid);
  if ((t & (kWave - 1)) == 0) {
    unsigned* slot = lds.reduce[t / kWave];
    slot[0] = low, slot[1] = ctz_all, slot[2] = ctz_val, slot[3] = first_mid;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    low = min(low, lds.reduce[w][0]);
    ctz_all = min(ctz_all, lds.reduce[w][1]);
    ctz_val = min(ctz_val, lds.reduce[w][2]);
    first_mid = min(first_mid, lds.reduce[w][3]);
  }
  // the same in every lane: keep them in scalar registers
  const int low_bit = static_cast<int>(__builtin_amdgcn_readfirstlane(low)) - 1024;
  const unsigned scale_ctz = __builtin_amdgcn_readfirstlane(ctz_all);
  const unsigned scale_value_ctz = __builtin_amdgcn_readfirstlane(ctz_val);
  const unsigned scale_first = __builtin_amdgcn_readfirstlane(first_mid);
  GridRound gr{};
  gr.scale_bits = rs.kind == DCQ_ROUND_F16 ? 11 : (rs.kind == DCQ_ROUND_BF16 || rs.kind == DCQ_ROUND_INT) ? 8
                  : scale_table ? 24 - static_cast<int>(scale_value_ctz) : 24;
  {
    const int low = low_bit;
    if (low >= -100 && low <= 100) {
      const float inv_half = ldexpf(1.0f, 1 - low);
      const float first = lds.tensor[0] * inv_half, last = lds.tensor[C - 1] * inv_half;
      // the clamp range must lie inside the table, or an index could fall outside the list
      const bool inside = rt.lo >= lds.tensor[0] && rt.hi <= lds.tensor[C - 1];
      if (inside && fabsf(first) < 0x1p22f && fabsf(last) < 0x1p22f && last - first <= static_cast<float>(kMaxValueCells)) {
        gr.inv_half = inv_half;
        gr.first = static_cast<int>(first);
        gr.cells = static_cast<int>(last - first);
      }
    }
  }
  if (scale_table && scale_first < 0x7f800000u && rs.hi <= lds.scale[CS - 1] && rs.hi > 0.0f) {
    const unsigned shift = scale_ctz;
    const int base = static_cast<int>(scale_first >> shift) - 1;  // cell 0: everything below the first midpoint
    const int cells = static_cast<int>(__float_as_uint(rs.hi) >> shift) - base;
    if (cells >= 1 && cells <= kMaxScaleCells) {
      gr.scale_shift = shift;
      gr.scale_base = base;
      gr.scale_cells = cells;
    }
  }
  if (gr.cells > 0) {
    const float half = 1.0f / gr.inv_half;  // exact
    for (int c = threadIdx.x; c <= gr.cells; c += blockDim.x)
      lds.value[c] = table_nearest(static_cast<float>(c + gr.first) * half, lds.tensor, C);
  }
  if (gr.scale_cells > 0) {
    for (int c = threadIdx.x; c <= gr.scale_cells; c += blockDim.x) {
      const float edge = c == 0 ? 0.0f : __uint_as_float(static_cast<unsigned>(c + gr.scale_base) << gr.scale_shift);
      lds.scale_value[c] = table_nearest(edge, lds.scale, CS);
    }
  }
  __syncthreads();
  return gr;
}

__device__ __forceinline__ float round_to_table(float t, const dcq_rounding& r, const GridRound& gr,
                                                const CodebookLut& lds) {
  float y;
  const float c = clamp_f(t, r.lo, r.hi);  // == fminf(fmaxf(t, lo), hi), NaN included (both give lo)
  if (gr.cells > 0) {
    y = lds.value[static_cast<int>(floorf(c * gr.inv_half)) - gr.first];
  } else {
    y = codebook_nearest(c, lds.tensor, r.codebook_size);
  }
  if (r.has_clamp) y = clamp_f(y, r.clamp_lo, r.clamp_hi);
  return y;
}

// apply_rounding of a magnitude (v >= 0) with the scale's dtype
__device__ __forceinline__ float round_scale_magnitude(float v, const dcq_rounding& r, const GridRound& gr,
                                                       const CodebookLut& lds) {
  if (r.kind != DCQ_ROUND_CODEBOOK || gr.scale_cells == 0) return apply_rounding(r, v, lds.scale);
  const float c = clamp_f(v, r.lo, r.hi);
  const int k = static_cast<int>(__float_as_uint(c) >> gr.scale_shift) - gr.scale_base;
  float y = lds.scale_value[k > 0 ? k : 0];
  if (r.has_clamp) y = clamp_f(y, r.clamp_lo, r.clamp_hi);
  return y;
}

// x / d.s for a divisor that is the same for every group (the quantization span, the per-tensor scale level)
__device__ __forceinline__ float div_uniform(float x, const Divisor& d) {
  return (d.fast && fabsf(x) <= 0x1p60f) ? div_fast(x, d) : x / d.s;
}

struct UniformDivisors {
  Divisor q_span, level0;
};

// Scale of one group from its largest magnitude (quantizer/impl/scale.py:46-62 for two levels).
__device__ __forceinline__ float table_group_scale(const CodebookParams& p, const CodebookLut& lds, const GridRound& gr,
                                                   const UniformDivisors& ud, float vmax, float level0, float& level1) {
  const dcq_step& st = p.step;
  if (p.has_ratio) vmax = vmax * p.ratio;
  if (st.has_bound_max) vmax = fminf(vmax, st.bound_max);
  float s;
  if (p.tensor_absmax) {
    level1 = round_scale_magnitude(div_uniform(fabsf(div_uniform(vmax, ud.q_span)), ud.level0), st.scale, gr, lds);
    s = level1 * level0;
  } else {
    s = round_scale_magnitude(fabsf(div_uniform(vmax, ud.q_span)), st.scale, gr, lds);
    level1 = s;
  }
  return s == 0.0f ? 1.0f : s;
}

// (int)floorf(x) in one instruction
__device__ __forceinline__ int floor_to_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// N values of one group held by one lane: quantize in place (v -> de-quantized), codes to q.  The list form has no
// branch per element: the caller has checked, for the whole wave, that the value list exists, that the table's own
// clamp is the only one and that every lane's divisor is inside the hoisted-reciprocal guard.
template <int N, bool SHORT>
__device__ __forceinline__ void quantize_by_list(float* v, float* q, float s, const Divisor& d, float lo, float hi,
                                                 float inv_half, const float* list) {
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    const f32x2 x2 = {v[i], v[i + 1]};
    const f32x2 t = SHORT ? div_short2(x2, d) : div_fast2(x2, d);
    // clamp (NaN -> lo, as fminf(fmaxf(t, lo), hi)), half-cell number (exact: a power-of-two factor, then floor)
    q[i] = list[floor_to_int(clamp_f(t.x, lo, hi) * inv_half)];
    q[i + 1] = list[floor_to_int(clamp_f(t.y, lo, hi) * inv_half)];
  }
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    const f32x2 o = f32x2{q[i], q[i + 1]} * f32x2{s, s};
    v[i] = o.x;
    v[i + 1] = o.y;
  }
}

template <int N>
__device__ __forceinline__ void quantize_to_table(float* v, float* q, float s, float local_abs, bool short_scale,
                                                  const dcq_rounding& r, const GridRound& gr, const CodebookLut& lds) {
  const Divisor d = make_divisor(s, local_abs);
  const bool by_list = gr.cells > 0 && !r.has_clamp && __builtin_amdgcn_ballot_w64(!d.fast) == 0;
  if (by_list) {
    const float* list = lds.value - gr.first;
    if (short_scale) {
      quantize_by_list<N, true>(v, q, s, d, r.lo, r.hi, gr.inv_half, list);
    } else {
      quantize_by_list<N, false>(v, q, s, d, r.lo, r.hi, gr.inv_half, list);
    }
    return;
  }
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    const float t = d.fast ? div_fast(v[i], d) : v[i] / s;
    q[i] = round_to_table(t, r, gr, lds);
    v[i] = q[i] * s;
  }
}

template <typename RAW>
__device__ __forceinline__ RAW swap_with_neighbour(const RAW& r) {
  RAW out;
  const uint32_t* in = reinterpret_cast<const uint32_t*>(&r);
  uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(in[i]), 0xB1, 0xf, 0xf, false));
  return out;
}

template <int ST, bool PAIRED>
__global__ __launch_bounds__(256) void fused_table_kernel(const CodebookParams p) {
  using E = Elem<ST>;
  using raw_t = typename E::raw_t;
  constexpr int VEC = E::kVec;
  __shared__ CodebookLut lds;
  const GridRound gr = build_lut(p, lds);
  const dcq_step& st = p.step;
  const raw_t* xin = static_cast<const raw_t*>(p.x);
  raw_t* dq_out = static_cast<raw_t*>(p.dequant);
  const bool nt = p.stream_hint != 0;
  // significand of a group scale: a value of the scale's dtype, times the per-tensor level when there are two (a
  // float8 level under a 16-bit one: 4 + 11 bits); up to 16 bits take the one-correction division
  const int level0_bits = !p.tensor_absmax ? 0 : p.level0.kind == DCQ_ROUND_F16 ? 11 : p.level0.kind == DCQ_ROUND_BF16 ? 8 : 24;
  const bool short_scale = gr.scale_bits + level0_bits <= 16;
  UniformDivisors ud;
  ud.q_span = make_divisor(st.q_span, 0.0f);

  float level0 = 1.0f;
  if (p.tensor_absmax) {
    float top = *p.tensor_absmax;
    if (p.has_ratio) top = top * p.ratio;
    if (st.has_bound_max) top = fminf(top, st.bound_max);
    level0 = apply_rounding(p.level0, fabsf(top / st.q_span) / p.level0_span, nullptr);
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.level0_out) *p.level0_out = level0;
  }
  ud.level0 = make_divisor(level0, 0.0f);

  float probe = 0.0f;
  if (PAIRED) {
    const int lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    // a workgroup owns `passes` consecutive runs of 4 waves x 128 pieces
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * p.passes) * 4 + (threadIdx.x >> 6);
    const int64_t pieces = p.G * 2;
    for (int pass = 0; pass < p.passes; ++pass) {
      const int64_t w = wave0 + static_cast<int64_t>(pass) * 4;
      if (w * 64 >= p.G) break;
      const int64_t r0 = w * 128 + lane, r1 = r0 + 64;
      raw_t a{}, b{};
      if (r0 < pieces) a = load_piece(xin + r0, nt);
      if (r1 < pieces) b = load_piece(xin + r1, nt);
      if (p.nonfinite) probe = E::probe(b, E::probe(a, probe));
      // even lanes give away their second piece, odd lanes their first
      const raw_t got = swap_with_neighbour(odd ? a : b);
      const raw_t first = odd ? got : a, second = odd ? b : got;
      const int64_t g = w * 64 + (odd ? 32 : 0) + (lane >> 1);
      float v[2 * VEC], q[2 * VEC];
      E::unpack(first, v);
      E::unpack(second, v + VEC);
      float mx = 0.0f;
#pragma unroll
      for (int i = 0; i < 2 * VEC; ++i) mx = fmaxf(mx, fabsf(v[i]));
      float level1;
      const float s = table_group_scale(p, lds, gr, ud, mx, level0, level1);
      quantize_to_table<2 * VEC>(v, q, s, mx, short_scale, st.tensor, gr, lds);
      if (g < p.G) {
        if (p.scale_out) p.scale_out[g] = s;
        if (p.zero_out) p.zero_out[g] = 0.0f;
        if (p.level1_out) p.level1_out[g] = level1;
        if (p.codes) {
          float4* co = reinterpret_cast<float4*>(p.codes + g * (2 * VEC));
#pragma unroll
          for (int i = 0; i < 2 * VEC / 4; ++i)
            store_piece(co + i, make_float4(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]), nt);
        }
      }
      if (dq_out) {
        const raw_t f = E::pack(v), sd = E::pack(v + VEC);
        const raw_t back = swap_with_neighbour(odd ? f : sd);
        const raw_t oa = odd ? back : f, ob = odd ? sd : back;
        if (r0 < pieces) store_piece(dq_out + r0, oa, nt);
        if (r1 < pieces) store_piece(dq_out + r1, ob, nt);
      }
    }
  } else {
    const int TL = p.team_lanes;
    const int tl = threadIdx.x & (TL - 1);
    const int64_t teams_per_pass = 256 >> p.team_shift;
    const int64_t team0 = static_cast<int64_t>(blockIdx.x) * p.passes * teams_per_pass + (threadIdx.x >> p.team_shift);
    for (int pass = 0; pass < p.passes; ++pass) {
      const int64_t g = team0 + pass * teams_per_pass;
      if (g >= p.G) break;  // (whole teams leave together)
      const int64_t at = g * TL + tl;
      float v[VEC], q[VEC];
      const raw_t piece = load_piece(xin + at, nt);
      if (p.nonfinite) probe = E::probe(piece, probe);
      E::unpack(piece, v);
      float mx = 0.0f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) mx = fmaxf(mx, fabsf(v[i]));
      const float vmax = team_reduce<true>(mx, TL);
      float level1;
      const float s = table_group_scale(p, lds, gr, ud, vmax, level0, level1);
      quantize_to_table<VEC>(v, q, s, mx, short_scale, st.tensor, gr, lds);
      if (tl == 0) {
        if (p.scale_out) p.scale_out[g] = s;
        if (p.zero_out) p.zero_out[g] = 0.0f;
        if (p.level1_out) p.level1_out[g] = level1;
      }
      if (p.codes) {
        float4* co = reinterpret_cast<float4*>(p.codes + at * VEC);
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i)
          store_piece(co + i, make_float4(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]), nt);
      }
      if (dq_out) store_piece(dq_out + at, E::pack(v), nt);
    }
  }
  report_nonfinite(probe, p.nonfinite);
}

template <int ST>
static int launch_table(CodebookParams p, bool paired, hipStream_t stream) {
  // Every workgroup builds the lookup lists (three barriers, a few hundred table searches) before it streams, so it
  // takes `passes` consecutive passes of 8 KiB of input -- one contiguous run -- to spread that cost; small tensors
  // keep one pass per workgroup so that the chip fills.
  // (measured on 32768 x 8192 bf16, NVFP4: 1 pass 387 us, 4 or 8 passes 230-235 us, 32 passes 246 us)
  const int64_t per_pass = paired ? 256 : (256 >> p.team_shift);  // groups per workgroup and pass
  const int64_t total = (p.G + per_pass - 1) / per_pass;
  int passes = 1;
  while (passes < 8 && total / (passes * 2) >= 4096) passes *= 2;
  p.passes = passes;
  const int64_t width = ST == DCQ_F32 ? 4 : 2;
  p.stream_hint = p.G * p.S * (width + (p.dequant ? width : 0) + (p.codes ? 4 : 0)) >= kStreamBytes;
  const int64_t blocks = (total + passes - 1) / passes;
  if (paired) {
    hipLaunchKernelGGL((fused_table_kernel<ST, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL((fused_table_kernel<ST, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  }
  DCQ_LAUNCH_CHECK("fused_table_kernel launch");
  return DCQ_OK;
}

// DCQ_ERR_UNSUPPORTED (no error text) when the configuration or the buffers need the general kernel.
int try_launch_fused_table(const dcq_step& st, const void* x, int storage, int64_t G, int64_t S, int has_ratio,
                           float ratio, const float* tensor_absmax, const dcq_rounding* level0, float level0_span,
                           void* dequant, float* codes, float* level0_out, float* level1_out, float* scale,
                           float* zero, hipStream_t stream, int32_t* nonfinite) {
  if (st.tensor.kind != DCQ_ROUND_CODEBOOK || st.zero_domain != DCQ_ZERO_NONE) return DCQ_ERR_UNSUPPORTED;
  const int vec = storage == DCQ_F32 ? 4 : 8;
  auto aligned = [](const void* ptr, uintptr_t a) { return ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) % a) == 0; };
  if (S % vec != 0 || !aligned(x, 16) || !aligned(dequant, 16) || !aligned(codes, 16)) return DCQ_ERR_UNSUPPORTED;
  const int64_t n = S / vec;
  const bool paired = n == 2;
  if (!paired && (n > 64 || (n & (n - 1)) != 0)) return DCQ_ERR_UNSUPPORTED;
  if (G >= (int64_t{1} << 38)) return DCQ_ERR_UNSUPPORTED;
  CodebookParams p{};
  p.step = st;
  p.x = x;
  p.dequant = dequant;
  p.codes = codes;
  p.scale_out = scale;
  p.zero_out = zero;
  p.G = G;
  p.S = static_cast<int32_t>(S);
  p.team_lanes = static_cast<int32_t>(n);
  p.team_shift = __builtin_ctz(static_cast<unsigned>(n));
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.tensor_absmax = tensor_absmax;
  if (tensor_absmax) {
    p.level0 = *level0;
    p.level0_span = level0_span;
  }
  p.level0_out = level0_out;
  p.level1_out = level1_out;
  p.nonfinite = nonfinite;
  switch (storage) {
    case DCQ_F32:
      return launch_table<DCQ_F32>(p, paired, stream);
    case DCQ_F16:
      return launch_table<DCQ_F16>(p, paired, stream);
    default:
      return launch_table<DCQ_BF16>(p, paired, stream);
  }
}

// =============================================================================================
// Smoothing candidates, fused (deepcompressor/calib/smooth.py:504-530): what the reference does to a weight as
//     w.mul(s) -> fake-quantize -> div(s) -> cast                    (mode 0)
// and to an activation as
//     x.div(s) -> fake-quantize -> mul(s) -> cast                    (mode 1)
// with s a per-channel scale, float32 in between, happens here in registers: one read of the tensor, and either one
// write of the processed tensor (one candidate) or no write at all -- then, for each of K candidate scale vectors, the
// group's error sum |processed - original|^degree evaluated in the tensor's dtype (the TensorError objective,
// calib/search.py:779-787), every candidate from the same registers.  Same team / chunk structure as the kernel above.
// =============================================================================================
struct SmoothParams {
  FusedParams f;
  const float* chan_scale;  // [K][C]
  int64_t C;
  int32_t groups_per_row;   // C / S
  int32_t divide_first;     // mode 1
  int32_t K;
  float* errors;            // [K][G] float32 sums (unrounded), or NULL
  const uint16_t* table;    // 16-bit storage: |d|^degree by bit pattern, or NULL when the degree is 2
  float degree;
  int8_t* codes;            // [K][G * S] codes minus zero point (the integer the scale multiplies), or NULL
  float* code_scales;       // [K][G] with codes
};

template <int ST>
__device__ __forceinline__ float error_term(float processed, float original, const SmoothParams& p) {
  // `processed` is already rounded to the storage type; difference, magnitude and power are each rounded to it too
  const float d = processed - original;
  if constexpr (ST == DCQ_F32) {
    const float m = fabsf(d);
    return p.degree == 2.0f ? m * m : powf(m, p.degree);
  } else {
    const float m = fabsf(Storage<ST>::round(d));
    if (p.table == nullptr) return Storage<ST>::round(m * m);
    const uint16_t bits = Storage<ST>::store(m) & 0x7fffu;
    return Storage<ST>::load(p.table[bits]);
  }
}

template <int ST, int ZD, int SK, bool SINGLE>
__global__ __launch_bounds__(256) void smooth_int_kernel(const SmoothParams sp) {
  using E = Elem<ST>;
  using raw_t = typename E::raw_t;
  constexpr int VEC = E::kVec;
  const FusedParams& p = sp.f;
  const int TL = p.team_lanes;
  const int tl = threadIdx.x & (TL - 1);
  const int64_t team0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> p.team_shift;
  const int64_t nteams = (static_cast<int64_t>(gridDim.x) * 256) >> p.team_shift;
  const raw_t* xin = static_cast<const raw_t*>(p.x);
  raw_t* out = static_cast<raw_t*>(p.dequant);
  const int64_t raws_per_group = p.S / VEC;
  const int chunks = SINGLE ? 1 : p.chunks;
  // whole waves iterate together (the trip count is taken from the wave's first team), dead teams redo group 0
  const int64_t wave_team0 = __shfl(team0, 0, kWave);
  float probe = 0.0f;
  for (int64_t gw = wave_team0, g0 = team0; gw < p.G; gw += nteams, g0 += nteams) {
    const bool live = g0 < p.G;
    const int64_t g = live ? g0 : 0;
    const raw_t* xg = xin + g * raws_per_group;
    const int64_t c_base = (g % sp.groups_per_row) * p.S;
    raw_t keep;
    if (SINGLE) {
      keep = xg[tl];
      if (live) probe = E::probe(keep, probe);
    } else if (live) {
      for (int c = 0; c < chunks; ++c) probe = E::probe(xg[static_cast<int64_t>(c) * TL + tl], probe);
    }
    for (int k = 0; k < sp.K; ++k) {
      const float* cs = sp.chan_scale + static_cast<int64_t>(k) * sp.C + c_base;
      // ---- pass 1: range of the smoothed values
      float mx = ZD == DCQ_ZERO_NONE ? 0.0f : -INFINITY, mn = INFINITY;
      for (int c = 0; c < chunks; ++c) {
        const int64_t off = static_cast<int64_t>(c) * TL + tl;
        float v[VEC], m[VEC];
        E::unpack(SINGLE ? keep : xg[off], v);
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
          const float4 m4 = *reinterpret_cast<const float4*>(cs + off * VEC + i);
          m[i] = m4.x, m[i + 1] = m4.y, m[i + 2] = m4.z, m[i + 3] = m4.w;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float xs = sp.divide_first ? v[i] / m[i] : v[i] * m[i];
          if (ZD == DCQ_ZERO_NONE) {
            mx = fmaxf(mx, fabsf(xs));
          } else {
            mx = fmaxf(mx, xs);
            mn = fminf(mn, xs);
          }
        }
      }
      const float local_abs = ZD == DCQ_ZERO_NONE ? mx : fmaxf(fabsf(mx), fabsf(mn));
      float vmax = team_reduce<true>(mx, TL), vmin = 0.0f;
      if (ZD != DCQ_ZERO_NONE) vmin = team_reduce<false>(mn, TL);
      if (p.has_ratio) {
        vmax = vmax * p.ratio;
        if (ZD != DCQ_ZERO_NONE) vmin = vmin * p.ratio;
      }
      if (ZD == DCQ_ZERO_PRE_SCALE) {
        vmax = fmaxf(vmax, 0.0f);
        vmin = fminf(vmin, 0.0f);
      }
      if (p.has_bound_max) vmax = fminf(vmax, p.bound_max);
      if (ZD != DCQ_ZERO_NONE && p.has_bound_min) vmin = fmaxf(vmin, p.bound_min);
      const float span = ZD != DCQ_ZERO_NONE ? vmax - vmin : vmax;
      float sc = round_scale<SK>(fabsf(span / p.q_span), 0.0f, 0.0f);
      if (sc == 0.0f) sc = 1.0f;
      float ze = 0.0f;
      if (ZD == DCQ_ZERO_PRE_SCALE) ze = fminf(fmaxf(rintf(p.q_min - vmin / sc), p.z_lo), p.z_hi);
      if (ZD == DCQ_ZERO_POST_SCALE) ze = round_scale<SK>(p.q_min * sc - vmin, 0.0f, 0.0f);
      const Divisor dv = make_divisor(sc, local_abs + fabsf(ze));
      // ---- pass 2: quantize, dequantize, take the smoothing back, cast
      float acc = 0.0f;
      for (int c = 0; c < chunks; ++c) {
        const int64_t off = static_cast<int64_t>(c) * TL + tl;
        float v[VEC], m[VEC], o[VEC], kept[VEC];
        E::unpack(SINGLE ? keep : xg[off], v);
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
          const float4 m4 = *reinterpret_cast<const float4*>(cs + off * VEC + i);
          m[i] = m4.x, m[i + 1] = m4.y, m[i + 2] = m4.z, m[i + 3] = m4.w;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float xs = sp.divide_first ? v[i] / m[i] : v[i] * m[i];
          float a = xs;
          if (ZD == DCQ_ZERO_POST_SCALE) a = a + ze;
          float t = dv.fast ? (short_significand(SK) ? div_short(a, dv) : div_fast(a, dv)) : a / sc;
          if (ZD == DCQ_ZERO_PRE_SCALE) t = t + ze;
          const float q = clamp_f(rintf(t), p.q_lo, p.q_hi);
          const float d = from_code<ZD>(q, sc, ze);
          o[i] = sp.divide_first ? d * m[i] : d / m[i];
          kept[i] = ZD == DCQ_ZERO_PRE_SCALE ? q - ze : q;
        }
        if (sp.codes) {
          if (live) {
            // the integer that the group's scale multiplies: code (minus a zero point in the code's own domain)
            uint32_t word[VEC / 4];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const uint32_t byte = static_cast<uint32_t>(static_cast<int>(kept[i])) & 0xffu;
              word[i / 4] = (i % 4 == 0) ? byte : word[i / 4] | (byte << (8 * (i % 4)));
            }
            uint32_t* dst = reinterpret_cast<uint32_t*>(sp.codes + (static_cast<int64_t>(k) * p.G + g) * p.S) +
                            off * (VEC / 4);
#pragma unroll
            for (int i = 0; i < VEC / 4; ++i) dst[i] = word[i];
          }
        } else if (sp.errors) {
          const raw_t packed = E::pack(o);  // rounds to the storage type exactly as the written tensor would be
          float r[VEC];
          E::unpack(packed, r);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc = acc + error_term<ST>(r[i], v[i], sp);
        } else if (live) {
          out[g * raws_per_group + off] = E::pack(o);
        }
      }
      if (sp.codes) {
        if (tl == 0 && live) sp.code_scales[static_cast<int64_t>(k) * p.G + g] = sc;
      } else if (sp.errors) {
        for (int w = TL >> 1; w > 0; w >>= 1) acc = acc + __shfl_xor(acc, w, kWave);
        if (tl == 0 && live) sp.errors[static_cast<int64_t>(k) * p.G + g] = acc;
      }
    }
  }
  report_nonfinite(probe, p.nonfinite);
}

template <int ST, int ZD, int SK>
static int launch_smooth(const SmoothParams& sp, hipStream_t stream) {
  const int64_t threads = sp.f.G * sp.f.team_lanes;
  int64_t blocks = (threads + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (sp.f.chunks == 1) {
    hipLaunchKernelGGL((smooth_int_kernel<ST, ZD, SK, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sp);
  } else {
    hipLaunchKernelGGL((smooth_int_kernel<ST, ZD, SK, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, sp);
  }
  DCQ_LAUNCH_CHECK("smooth_int_kernel launch");
  return DCQ_OK;
}

template <int ST, int ZD>
static int smooth_scale_kind(const SmoothParams& sp, int sk, hipStream_t stream) {
  switch (sk) {
    case DCQ_ROUND_F16:
      return launch_smooth<ST, ZD, DCQ_ROUND_F16>(sp, stream);
    case DCQ_ROUND_BF16:
      return launch_smooth<ST, ZD, DCQ_ROUND_BF16>(sp, stream);
    default:
      return launch_smooth<ST, ZD, DCQ_ROUND_F32>(sp, stream);
  }
}

template <int ST>
static int smooth_zero(const SmoothParams& sp, int zd, int sk, hipStream_t stream) {
  switch (zd) {
    case DCQ_ZERO_NONE:
      return smooth_scale_kind<ST, DCQ_ZERO_NONE>(sp, sk, stream);
    case DCQ_ZERO_PRE_SCALE:
      return smooth_scale_kind<ST, DCQ_ZERO_PRE_SCALE>(sp, sk, stream);
    default:
      return smooth_scale_kind<ST, DCQ_ZERO_POST_SCALE>(sp, sk, stream);
  }
}

__global__ void selftest_division_kernel(const float* x, const float* sv, float* fast, float* ref, int* used_fast,
                                         float* one_step, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Divisor d = make_divisor(sv[i], fabsf(x[i]));
  const f32x2 a = {x[i], x[i]};
  fast[i] = d.fast ? div_fast2(a, d).x : x[i] / sv[i];
  if (one_step) one_step[i] = d.fast ? div_short2(a, d).y : x[i] / sv[i];
  ref[i] = x[i] / sv[i];
  used_fast[i] = d.fast ? 1 : 0;
}

}  // namespace dcq

extern "C" int dcq_selftest_division(const float* x, const float* s, float* fast, float* ref, int32_t* used_fast,
                                     float* one_step, int64_t n, dcq_stream_t stream) {
  if (n <= 0) return DCQ_OK;
  hipLaunchKernelGGL(dcq::selftest_division_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, s, fast, ref, used_fast, one_step, n);
  DCQ_LAUNCH_CHECK("selftest_division_kernel launch");
  return DCQ_OK;
}

namespace dcq {
// the three outputs of the smoothing kernel: the processed tensor (one candidate), the error table, or codes + scales
static int smooth_entry(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                        int64_t groups_per_row, const float* chan_scale, int32_t K, int32_t mode, int32_t has_ratio,
                        float ratio, void* dequant, float* errors, double degree, const uint16_t* pow_table,
                        int8_t* codes, float* code_scales, int32_t* nonfinite, dcq_stream_t stream);
}  // namespace dcq

extern "C" int dcq_smooth_fake_quant_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                                            int64_t groups_per_row, const float* chan_scale, int32_t K, int32_t mode,
                                            int32_t has_ratio, float ratio, void* dequant, float* errors, double degree,
                                            const uint16_t* pow_table, int32_t* nonfinite, dcq_stream_t stream) {
  DCQ_REQUIRE((dequant != nullptr) != (errors != nullptr), DCQ_ERR_ARGUMENT,
              "exactly one of the processed tensor and the error table is written");
  DCQ_REQUIRE(dequant == nullptr || K == 1, DCQ_ERR_ARGUMENT, "the processed tensor is written for one candidate");
  return dcq::smooth_entry(step, x, storage, G, S, groups_per_row, chan_scale, K, mode, has_ratio, ratio, dequant, errors,
                           degree, pow_table, nullptr, nullptr, nonfinite, stream);
}

extern "C" int dcq_smooth_quantize_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                                          int64_t groups_per_row, const float* chan_scale, int32_t K, int32_t mode,
                                          int32_t has_ratio, float ratio, int8_t* codes, float* scales,
                                          int32_t* nonfinite, dcq_stream_t stream) {
  DCQ_REQUIRE(G == 0 || (codes != nullptr && scales != nullptr), DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(step != nullptr, DCQ_ERR_ARGUMENT, "step is NULL");
  // the stored byte is the integer the scale multiplies: it has to fit, and a zero point has to live in its domain
  DCQ_REQUIRE(step->zero_domain != DCQ_ZERO_POST_SCALE, DCQ_ERR_UNSUPPORTED,
              "a zero point subtracted after the scale leaves no integer code");
  const bool pre = step->zero_domain == DCQ_ZERO_PRE_SCALE;
  const float top = step->tensor.hi - (pre ? step->zero.lo : 0.0f), bottom = step->tensor.lo - (pre ? step->zero.hi : 0.0f);
  DCQ_REQUIRE(top <= 127.0f && bottom >= -128.0f, DCQ_ERR_UNSUPPORTED, "code minus zero point does not fit a signed byte");
  return dcq::smooth_entry(step, x, storage, G, S, groups_per_row, chan_scale, K, mode, has_ratio, ratio, nullptr, nullptr,
                           2.0, nullptr, codes, scales, nonfinite, stream);
}

namespace dcq {
static int smooth_entry(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                        int64_t groups_per_row, const float* chan_scale, int32_t K, int32_t mode, int32_t has_ratio,
                        float ratio, void* dequant, float* errors, double degree, const uint16_t* pow_table,
                        int8_t* codes, float* code_scales, int32_t* nonfinite, dcq_stream_t stream) {
  DCQ_REQUIRE(step != nullptr, DCQ_ERR_ARGUMENT, "step is NULL");
  DCQ_REQUIRE(G >= 0 && S > 0 && groups_per_row > 0 && K > 0, DCQ_ERR_ARGUMENT, "bad sizes");
  DCQ_REQUIRE(mode == 0 || mode == 1, DCQ_ERR_ARGUMENT, "mode must be 0 (multiply first) or 1 (divide first)");
  if (G == 0) return DCQ_OK;
  DCQ_REQUIRE(x && chan_scale, DCQ_ERR_ARGUMENT, "NULL buffer");
  const dcq_step& st = *step;
  const int sk = st.scale.kind;
  const bool scale_ok = (sk == DCQ_ROUND_F32 || sk == DCQ_ROUND_F16 || sk == DCQ_ROUND_BF16) && !st.scale.has_clamp;
  bool zero_ok = true;
  if (st.zero_domain == DCQ_ZERO_PRE_SCALE) zero_ok = st.zero.kind == DCQ_ROUND_INT;
  if (st.zero_domain == DCQ_ZERO_POST_SCALE) zero_ok = st.zero.kind == sk && !st.zero.has_clamp;
  DCQ_REQUIRE(st.tensor.kind == DCQ_ROUND_INT && scale_ok && zero_ok, DCQ_ERR_UNSUPPORTED,
              "the fused smoothing kernel takes integer types with one float scale level");
  const int vec = storage == DCQ_F32 ? 4 : 8;
  auto aligned = [](const void* ptr, uintptr_t a) { return ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) % a) == 0; };
  DCQ_REQUIRE(S % vec == 0 && S <= (1 << 30) && aligned(x, 16) && aligned(dequant, 16) && aligned(chan_scale, 16),
              DCQ_ERR_UNSUPPORTED, "groups must be whole 16-byte pieces of 16-byte aligned buffers");
  const int64_t n = S / vec;
  int lanes = 1;
  while (lanes < 64 && n % (lanes * 2) == 0) lanes *= 2;
  SmoothParams sp{};
  FusedParams& p = sp.f;
  p.x = x;
  p.dequant = dequant;
  p.G = G;
  p.S = static_cast<int32_t>(S);
  p.team_lanes = lanes;
  p.team_shift = __builtin_ctz(static_cast<unsigned>(lanes));
  p.chunks = static_cast<int32_t>(n / lanes);
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.q_lo = st.tensor.lo;
  p.q_hi = st.tensor.hi;
  p.z_lo = st.zero.lo;
  p.z_hi = st.zero.hi;
  p.q_span = st.q_span;
  p.q_min = st.q_min;
  p.has_bound_min = st.has_bound_min;
  p.has_bound_max = st.has_bound_max;
  p.bound_min = st.bound_min;
  p.bound_max = st.bound_max;
  p.nonfinite = nonfinite;
  sp.chan_scale = chan_scale;
  sp.groups_per_row = static_cast<int32_t>(groups_per_row);
  sp.C = groups_per_row * S;
  sp.divide_first = mode;
  sp.K = K;
  sp.errors = errors;
  sp.codes = codes;
  sp.code_scales = code_scales;
  sp.degree = static_cast<float>(degree);
  sp.table = nullptr;
  if (errors != nullptr && storage != DCQ_F32 && round_degree_to_storage(degree, storage) != 2.0) {
    DCQ_REQUIRE(pow_table != nullptr, DCQ_ERR_ARGUMENT, "pow_table is NULL (build it with dcq_pow_table)");
    sp.table = pow_table;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (storage) {
    case DCQ_F32:
      return smooth_zero<DCQ_F32>(sp, st.zero_domain, sk, s);
    case DCQ_F16:
      return smooth_zero<DCQ_F16>(sp, st.zero_domain, sk, s);
    case DCQ_BF16:
      return smooth_zero<DCQ_BF16>(sp, st.zero_domain, sk, s);
    default:
      set_error("bad storage %d", storage);
      return DCQ_ERR_ARGUMENT;
  }
}
}  // namespace dcq
