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
  float z_lo, z_hi;     // zero range (PreScale)
  float q_span, q_min;
  int32_t has_bound_min, has_bound_max;
  float bound_min, bound_max;
};

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
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = static_cast<uint32_t>(bf16_rne_hw(v[2 * i])) | (static_cast<uint32_t>(bf16_rne_hw(v[2 * i + 1])) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
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
};

// xor-butterfly max / min over a team of TL adjacent lanes.  The first four steps stay inside a 16-lane row
// and are DPP operand modifiers (no LDS round trip): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then -- every
// quad being uniform by now -- row_half_mirror pairs the two quads of a half row and row_mirror the two halves.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <bool IS_MAX>
__device__ __forceinline__ float team_reduce(float v, int TL) {
  auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); };
  if (TL >= 2) v = op(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
  if (TL >= 4) v = op(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
  if (TL >= 8) v = op(v, dpp_move<0x141>(v));  // row_half_mirror
  if (TL >= 16) v = op(v, dpp_move<0x140>(v));  // row_mirror
  if (TL >= 32) v = op(v, __shfl_xor(v, 16, kWave));
  if (TL >= 64) v = op(v, __shfl_xor(v, 32, kWave));
  return v;
}

__device__ __forceinline__ float clamp_f(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

template <int SK>
__device__ __forceinline__ float round_scale(float v) {
  if (SK == DCQ_ROUND_F16) return f16_to_f32(f32_to_f16(v));
  if (SK == DCQ_ROUND_BF16) return __uint_as_float(static_cast<uint32_t>(bf16_rne_hw(v)) << 16);
  return v;
}

// groups a team of lanes holds at once in the one-chunk case; measured on MI355X (uint4 + zero point, groups of 128,
// bf16): 1 -> 5.5-5.6 TB/s, 2 -> 5.2, 4 -> 4.5 on 14336 x 4096
#ifndef DCQ_FUSED_U
#define DCQ_FUSED_U 1
#endif

template <int ST, int ZD, int SK, bool SINGLE>
__global__ __launch_bounds__(256) void fused_int_kernel(const FusedParams p) {
  using E = Elem<ST>;
  using raw_t = typename E::raw_t;
  constexpr int VEC = E::kVec;
  const int TL = p.team_lanes;
  const int tl = threadIdx.x & (TL - 1);
  const int64_t team0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> p.team_shift;
  const int64_t nteams = (static_cast<int64_t>(gridDim.x) * 256) >> p.team_shift;
  const raw_t* xin = static_cast<const raw_t*>(p.x);
  raw_t* dq_out = static_cast<raw_t*>(p.dequant);
  const int64_t raws_per_group = p.S / VEC;

  constexpr int U = SINGLE ? DCQ_FUSED_U : 1;  // groups in flight per team: more bytes outstanding per lane
  for (int64_t g0 = team0; g0 < p.G; g0 += nteams * U) {
    raw_t keep[U];
    float s[U], z[U];
    Divisor dv[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = g0 + u * nteams;
      live[u] = g < p.G;
      if (SINGLE && live[u]) keep[u] = xin[g * raws_per_group + tl];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = live[u] ? g0 + u * nteams : g0;
      const raw_t* xg = xin + g * raws_per_group;
      float vmax, vmin = 0.0f;
      float local_abs = INFINITY;  // bound on |x| over the elements THIS lane will quantize
      if (p.range_max == nullptr) {
        float mx = ZD == DCQ_ZERO_NONE ? 0.0f : -INFINITY, mn = INFINITY;
        for (int c = 0; c < (SINGLE ? 1 : p.chunks); ++c) {
          float v[VEC];
          if (SINGLE) {
            E::unpack(live[u] ? keep[u] : keep[0], v);
          } else {
            E::unpack(xg[static_cast<int64_t>(c) * TL + tl], v);
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
          float v[VEC];
          E::unpack(live[u] ? keep[u] : keep[0], v);
          local_abs = 0.0f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) local_abs = fmaxf(local_abs, fabsf(v[i]));
        }
      }
      if (p.has_bound_max) vmax = fminf(vmax, p.bound_max);
      if (ZD != DCQ_ZERO_NONE && p.has_bound_min) vmin = fmaxf(vmin, p.bound_min);
      const float span = ZD != DCQ_ZERO_NONE ? vmax - vmin : vmax;
      float sc = round_scale<SK>(fabsf(span / p.q_span));
      if (sc == 0.0f) sc = 1.0f;
      float ze = 0.0f;
      if (ZD == DCQ_ZERO_PRE_SCALE) ze = fminf(fmaxf(rintf(p.q_min - vmin / sc), p.z_lo), p.z_hi);
      if (ZD == DCQ_ZERO_POST_SCALE) ze = round_scale<SK>(p.q_min * sc - vmin);
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
      for (int c = 0; c < (SINGLE ? 1 : p.chunks); ++c) {
        const int64_t off = static_cast<int64_t>(c) * TL + tl;
        const raw_t raw = SINGLE ? keep[u] : xg[off];
        float v[VEC], d[VEC], qv[VEC];
        E::unpack(raw, v);
        if (dv[u].fast) {
#pragma unroll
          for (int i = 0; i < VEC; i += 2) {
            f32x2 a = {v[i], v[i + 1]};
            const f32x2 zz = {z[u], z[u]}, ss = {s[u], s[u]};
            if (ZD == DCQ_ZERO_POST_SCALE) a = a + zz;
            f32x2 t = div_fast2(a, dv[u]);
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
        if (dq_out) dq_out[g * raws_per_group + off] = E::pack(d);
        if (p.codes) {
          float4* co = reinterpret_cast<float4*>(p.codes + (g * raws_per_group + off) * VEC);
#pragma unroll
          for (int i = 0; i < VEC / 4; ++i)
            co[i] = make_float4(qv[4 * i], qv[4 * i + 1], qv[4 * i + 2], qv[4 * i + 3]);
        }
      }
    }
  }
}

template <int ST, int ZD, int SK>
static int launch_fused(const FusedParams& p, hipStream_t stream) {
  const int64_t threads = p.G * p.team_lanes;
  int64_t blocks = (threads + 255) / 256;
  if (p.chunks == 1) blocks = (blocks + DCQ_FUSED_U - 1) / DCQ_FUSED_U;  // several groups per team and pass
  // measured on MI355X: a persistent grid of 8192 workgroups (32 per CU), each team looping over its groups, streams
  // fastest from 60 MB to 1 GB (5.5-5.6 TB/s at 235 MB, 4.5-4.7 TB/s beyond the Infinity Cache); 2048 workgroups are
  // ~8 % slower, an un-looped grid 3-8 %; non-temporal loads / stores change nothing
  constexpr int64_t kGridCap = 8192;
  if (blocks > kGridCap) blocks = kGridCap;
  if (p.chunks == 1) {
    hipLaunchKernelGGL((fused_int_kernel<ST, ZD, SK, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,
                       p);
  } else {
    hipLaunchKernelGGL((fused_int_kernel<ST, ZD, SK, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,
                       p);
  }
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
                         float* zero, hipStream_t stream) {
  const int sk = st.scale.kind;
  const bool scale_ok = (sk == DCQ_ROUND_F32 || sk == DCQ_ROUND_F16 || sk == DCQ_ROUND_BF16) && !st.scale.has_clamp;
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
  p.q_span = st.q_span;
  p.q_min = st.q_min;
  p.has_bound_min = st.has_bound_min;
  p.has_bound_max = st.has_bound_max;
  p.bound_min = st.bound_min;
  p.bound_max = st.bound_max;
  switch (storage) {
    case DCQ_F32:
      return fused_zero<DCQ_F32>(p, st.zero_domain, sk, stream);
    case DCQ_F16:
      return fused_zero<DCQ_F16>(p, st.zero_domain, sk, stream);
    default:
      return fused_zero<DCQ_BF16>(p, st.zero_domain, sk, stream);
  }
}

__global__ void selftest_division_kernel(const float* x, const float* sv, float* fast, float* ref, int* used_fast,
                                         int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Divisor d = make_divisor(sv[i], fabsf(x[i]));
  const f32x2 a = {x[i], x[i]};
  fast[i] = d.fast ? div_fast2(a, d).x : x[i] / sv[i];
  ref[i] = x[i] / sv[i];
  used_fast[i] = d.fast ? 1 : 0;
}

}  // namespace dcq

extern "C" int dcq_selftest_division(const float* x, const float* s, float* fast, float* ref, int32_t* used_fast,
                                     int64_t n, dcq_stream_t stream) {
  if (n <= 0) return DCQ_OK;
  hipLaunchKernelGGL(dcq::selftest_division_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, s, fast, ref, used_fast, n);
  DCQ_LAUNCH_CHECK("selftest_division_kernel launch");
  return DCQ_OK;
}
