// Fake-quantization kernels on canonical groups x[G][S] (gfx950).
//
// One "team" of TL adjacent lanes (TL a power of two, 1..64) owns one group at a time: every lane
// moves 16 bytes per access, so a wave always issues fully coalesced 1 KiB loads/stores, the
// group's min/max is an xor-butterfly inside the team, and scale/zero live in registers.  The
// kernel is HBM-bound: one read and one write of the tensor (groups longer than one team pass are
// re-read from L2 for the second pass).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "dcq_common.cuh"

namespace dcq {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int hip_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return DCQ_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return DCQ_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// kernel parameters
// ---------------------------------------------------------------------------------------------
enum GroupMode { kFused = 0, kRangeOnly = 1, kQuantOnly = 2, kDequantOnly = 3 };

struct GroupParams {
  dcq_step step;
  const void* x;           // input (storage type; float codes for kDequantOnly)
  int64_t G, S;
  int32_t team_lanes;      // TL
  int32_t chunks;          // S / (TL * VEC)
  // range inputs
  const float* range_max;  // static range (fused) or NULL
  const float* range_min;
  int32_t has_ratio;
  float ratio;
  // given scale/zero (quant-only / dequant-only)
  const float* scale_in;
  const float* zero_in;
  const float* round_delta;  // [G][S]: codes = clamp(floor(t) + delta) instead of round-to-nearest, or NULL
  // two scale levels, the first one per tensor (NVFP4-style): the largest un-quantized group scale of the whole
  // tensor (device scalar, produced by tensor_absmax_kernel), the first level's rounding and span; the second level
  // is step.scale.  Outputs: level0_out[1], level1_out[G] (scale_out then receives the compound).
  const float* tensor_absmax;
  dcq_rounding level0;
  float level0_span;
  float* level0_out;
  float* level1_out;
  // previous progressive step
  const float* outer;
  float outer_lo, outer_hi;
  int32_t deq_clamp;
  float deq_lo, deq_hi;
  // outputs
  void* dequant;
  float* codes;
  float* scale_out;
  float* zero_out;
  float* vmax_out;
  float* vmin_out;
  int32_t* nonfinite;  // |= 4 when an element of x is NaN or +-Inf (NULL: not wanted)
};

constexpr int kBlock = 256;

// Codebooks (<= 256 entries each) are staged in LDS once per block: [0] tensor, [1] scale, [2] zero.
struct CodebookLds {
  float v[3][256];
};

__device__ __forceinline__ void stage_codebooks(const dcq_step& st, CodebookLds& lds) {
  const dcq_rounding* r[3] = {&st.tensor, &st.scale, &st.zero};
  bool any = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (r[k]->kind == DCQ_ROUND_CODEBOOK) {
      any = true;
      for (int i = threadIdx.x; i < r[k]->codebook_size; i += blockDim.x) lds.v[k][i] = r[k]->codebook[i];
    }
  }
  if (any) __syncthreads();
}

template <int ST, int VEC, int ZD, int MODE, int OUT_ST, bool SINGLE>
__global__ __launch_bounds__(kBlock) void group_kernel(const GroupParams p) {
  using In = Storage<(MODE == kDequantOnly) ? DCQ_F32 : ST>;
  using Out = Storage<OUT_ST>;
  using in_t = typename In::elem_t;
  using out_t = typename Out::elem_t;
  __shared__ CodebookLds cbs;
  stage_codebooks(p.step, cbs);
  const dcq_step& st = p.step;

  const int TL = p.team_lanes;
  const int tl = threadIdx.x & (TL - 1);
  const int64_t team0 = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / TL;
  const int64_t nteams = static_cast<int64_t>(gridDim.x) * kBlock / TL;
  const in_t* xin = static_cast<const in_t*>(p.x);
  const bool want_q = (MODE == kFused || MODE == kQuantOnly);

  for (int64_t g = team0; g < p.G; g += nteams) {
    const in_t* xg = xin + g * p.S;
    const float outer = p.outer ? p.outer[g] : 1.0f;
    // x / outer for many x: the hoisted-reciprocal form of the IEEE division (bit-identical inside its guard, which
    // is checked per element on |x|; see make_divisor)
    const Divisor d_outer = make_divisor(outer, 0.0f);
    auto by_outer = [&](float v) {
      return (d_outer.fast && fabsf(v) <= 0x1p60f) ? div_fast(v, d_outer) : v / outer;
    };
    Chunk<in_t, VEC> keep;  // the group's data when it fits one pass
    float s = 1.0f, z = 0.0f;
    float probe = 0.0f;  // stays 0 unless an element is NaN or +-Inf (x * 0 is NaN then)

    if (MODE == kFused || MODE == kRangeOnly) {
      float vmax, vmin = 0.0f;
      const bool measure = (MODE == kRangeOnly) || p.range_max == nullptr;
      if (measure) {
        float mx = (ZD == DCQ_ZERO_NONE) ? 0.0f : -INFINITY, mn = INFINITY;
        for (int c = 0; c < p.chunks; ++c) {
          const Chunk<in_t, VEC> ch = load_chunk<in_t, VEC>(xg + (static_cast<int64_t>(c) * TL + tl) * VEC);
          if (SINGLE) keep = ch;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            float v = In::load(ch.v[i]);
            if (MODE == kRangeOnly) probe = __builtin_fmaf(v, 0.0f, probe);
            if (p.outer) v = fminf(fmaxf(by_outer(v), p.outer_lo), p.outer_hi);
            if (ZD == DCQ_ZERO_NONE) {
              mx = fmaxf(mx, fabsf(v));
            } else {
              mx = fmaxf(mx, v);
              mn = fminf(mn, v);
            }
          }
        }
        vmax = team_max(mx, TL);
        if (ZD != DCQ_ZERO_NONE) vmin = team_min(mn, TL);
        if (MODE == kRangeOnly) {
          if (tl == 0) {
            p.vmax_out[g] = vmax;
            if (ZD != DCQ_ZERO_NONE) p.vmin_out[g] = vmin;
          }
          if (p.nonfinite != nullptr && probe != probe) atomicOr(p.nonfinite, 4);
          continue;
        }
        // DynamicRange.measure: ratio, then the PreScale sign clamp
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
        if (SINGLE) keep = load_chunk<in_t, VEC>(xg + static_cast<int64_t>(tl) * VEC);
      }
      if (p.tensor_absmax) {
        // quantize_scale with two levels (quantizer/impl/scale.py:46-62): the first level quantizes the tensor's
        // largest group scale over the second level's span, the second level each group's scale relative to it.
        // Division by q_span is monotone, so the largest group scale is the scale of the largest magnitude.
        float top = *p.tensor_absmax;
        if (p.has_ratio) top = top * p.ratio;
        if (st.has_bound_max) {
          top = fminf(top, st.bound_max);
          vmax = fminf(vmax, st.bound_max);
        }
        const float level0 = apply_rounding(p.level0, fabsf(top / st.q_span) / p.level0_span, nullptr);
        const float level1 = apply_rounding(st.scale, fabsf(vmax / st.q_span) / level0, cbs.v[1]);
        s = level1 * level0;
        if (s == 0.0f) s = 1.0f;
        if (tl == 0) {
          if (p.level1_out) p.level1_out[g] = level1;
          if (g == 0 && p.level0_out) *p.level0_out = level0;
        }
      } else {
        const ScaleZero sz = scale_zero_from_range(st, vmax, vmin, cbs.v[1], cbs.v[2]);
        s = sz.s;
        z = sz.z;
      }
      if (tl == 0) {
        if (p.scale_out) p.scale_out[g] = s;
        if (p.zero_out) p.zero_out[g] = z;
      }
    } else {
      s = p.scale_in[g];
      if (ZD != DCQ_ZERO_NONE) z = p.zero_in[g];
    }

    // second pass: codes and dequantized values
    const Divisor d_scale = make_divisor(s, 0.0f);
    for (int c = 0; c < p.chunks; ++c) {
      const int64_t off = (static_cast<int64_t>(c) * TL + tl) * VEC;
      Chunk<in_t, VEC> ch;
      if (SINGLE && MODE == kFused) {
        ch = keep;
      } else {
        ch = load_chunk<in_t, VEC>(xg + off);
      }
      Chunk<out_t, VEC> dq;
      Chunk<float, VEC> qc, delta;
      if (want_q && p.round_delta) delta = load_chunk<float, VEC>(p.round_delta + g * p.S + off);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float q;
        if (want_q) {
          float v = In::load(ch.v[i]);
          probe = __builtin_fmaf(v, 0.0f, probe);
          if (p.outer) v = fminf(fmaxf(by_outer(v), p.outer_lo), p.outer_hi);
          float t = v;
          if (ZD == DCQ_ZERO_POST_SCALE) t = t + z;
          t = (d_scale.fast && fabsf(t) <= 0x1p60f) ? div_fast(t, d_scale) : t / s;
          if (ZD == DCQ_ZERO_PRE_SCALE) t = t + z;
          if (p.round_delta) {  // integer types only (checked on the host): simple.py:47-51
            q = fminf(fmaxf(floorf(t) + delta.v[i], st.tensor.lo), st.tensor.hi);
          } else {
            q = (st.tensor.kind == DCQ_ROUND_INT) ? round_int(t, st.tensor.lo, st.tensor.hi)
                                                   : apply_rounding(st.tensor, t, cbs.v[0]);
          }
        } else {
          q = ch.v[i];
        }
        qc.v[i] = q;
        float d = from_code<ZD>(q, s, z);
        if (p.outer) {
          if (p.deq_clamp) d = fminf(fmaxf(d, p.deq_lo), p.deq_hi);
          d = d * outer;
        }
        dq.v[i] = Out::store(d);
      }
      if (p.dequant) store_chunk<out_t, VEC>(static_cast<out_t*>(p.dequant) + g * p.S + off, dq);
      if (p.codes) store_chunk<float, VEC>(p.codes + g * p.S + off, qc);
    }
    if (p.nonfinite != nullptr && probe != probe) atomicOr(p.nonfinite, 4);
  }
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
static int storage_size(int st) { return st == DCQ_F32 ? 4 : 2; }

struct TeamPlan {
  int vec, lanes, chunks;
};

// `vec` elements per lane and access (all of x, dequant and codes must allow it, else 1), then the
// widest power-of-two team whose passes tile the group exactly.
static TeamPlan plan_team(int64_t S, int vec, const void* x, int x_bytes, const void* dq, int dq_bytes,
                          const void* codes) {
  auto ok = [&](const void* p, int bytes) {
    return p == nullptr || (reinterpret_cast<uintptr_t>(p) % static_cast<uintptr_t>(vec * bytes)) == 0;
  };
  if (S % vec != 0 || !ok(x, x_bytes) || !ok(dq, dq_bytes) || !ok(codes, 4)) vec = 1;
  const int64_t n = S / vec;
  int lanes = 1;
  while (lanes < 64 && n % (lanes * 2) == 0) lanes *= 2;
  return {vec, lanes, static_cast<int>(n / lanes)};
}

static int validate_step(const dcq_step* st) {
  DCQ_REQUIRE(st != nullptr, DCQ_ERR_ARGUMENT, "step is NULL");
  DCQ_REQUIRE(st->zero_domain >= DCQ_ZERO_NONE && st->zero_domain <= DCQ_ZERO_POST_SCALE, DCQ_ERR_ARGUMENT,
              "bad zero_domain %d", st->zero_domain);
  const dcq_rounding* r[3] = {&st->tensor, &st->scale, &st->zero};
  for (int k = 0; k < 3; ++k) {
    DCQ_REQUIRE(r[k]->kind >= DCQ_ROUND_F32 && r[k]->kind <= DCQ_ROUND_EXP, DCQ_ERR_ARGUMENT, "bad rounding kind %d",
                r[k]->kind);
    if (r[k]->kind == DCQ_ROUND_CODEBOOK) {
      DCQ_REQUIRE(r[k]->codebook != nullptr && r[k]->codebook_size > 0 && r[k]->codebook_size <= 256, DCQ_ERR_ARGUMENT,
                  "codebook rounding needs 1..256 device values (got %d)", r[k]->codebook_size);
    }
  }
  DCQ_REQUIRE(st->tensor.kind == DCQ_ROUND_INT || st->tensor.kind == DCQ_ROUND_CODEBOOK, DCQ_ERR_UNSUPPORTED,
              "tensor dtype must be an integer or a codebook (floating-point) type");
  return DCQ_OK;
}

// A floating-point type with a zero point takes a range centred on the group mean (data/range.py:323-341), which
// the kernels that measure the range themselves do not compute: such a step needs its range from the caller.
static bool needs_mean_centred_range(const dcq_step* st) {
  return st->tensor.kind == DCQ_ROUND_CODEBOOK && st->zero_domain != DCQ_ZERO_NONE;
}

template <int ST, int VEC, int ZD, int MODE, int OUT_ST>
static int launch_group(const GroupParams& p, hipStream_t stream) {
  const int64_t threads = p.G * p.team_lanes;
  int64_t blocks = (threads + kBlock - 1) / kBlock;
  const int64_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (p.chunks == 1) {
    hipLaunchKernelGGL((group_kernel<ST, VEC, ZD, MODE, OUT_ST, true>), dim3(static_cast<unsigned>(blocks)),
                       dim3(kBlock), 0, stream, p);
  } else {
    hipLaunchKernelGGL((group_kernel<ST, VEC, ZD, MODE, OUT_ST, false>), dim3(static_cast<unsigned>(blocks)),
                       dim3(kBlock), 0, stream, p);
  }
  DCQ_LAUNCH_CHECK("group_kernel launch");
  return DCQ_OK;
}

template <int ST, int VEC, int MODE, int OUT_ST>
static int dispatch_zero(const GroupParams& p, hipStream_t stream) {
  switch (p.step.zero_domain) {
    case DCQ_ZERO_NONE:
      return launch_group<ST, VEC, DCQ_ZERO_NONE, MODE, OUT_ST>(p, stream);
    case DCQ_ZERO_PRE_SCALE:
      return launch_group<ST, VEC, DCQ_ZERO_PRE_SCALE, MODE, OUT_ST>(p, stream);
    default:
      return launch_group<ST, VEC, DCQ_ZERO_POST_SCALE, MODE, OUT_ST>(p, stream);
  }
}

template <int ST, int MODE, int OUT_ST>
static int dispatch_vec(const GroupParams& p, int vec, hipStream_t stream) {
  constexpr int kFull = Storage<(MODE == kDequantOnly) ? DCQ_F32 : ST>::kVec;
  // the output chunk must also be a single 16-byte (or narrower) access: use the narrower of the two widths
  constexpr int kOutFull = Storage<OUT_ST>::kVec;
  constexpr int kV = kFull < kOutFull ? kFull : kOutFull;
  if (vec >= kV) return dispatch_zero<ST, kV, MODE, OUT_ST>(p, stream);
  return dispatch_zero<ST, 1, MODE, OUT_ST>(p, stream);
}

// in/out storage combinations that occur: same type, or float32 working data written back as 16-bit
template <int MODE>
static int dispatch_storage(GroupParams& p, int storage, int out_storage, hipStream_t stream) {
  const int in_bytes = (MODE == kDequantOnly) ? 4 : storage_size(storage);
  const int out_bytes = storage_size(out_storage);
  const int wide = in_bytes > out_bytes ? in_bytes : out_bytes;
  const TeamPlan tp = plan_team(p.S, 16 / wide, p.x, in_bytes, p.dequant, out_bytes, p.codes);
  p.team_lanes = tp.lanes;
  p.chunks = tp.chunks;
#define DCQ_CASE(IN, OUT) \
  if (storage == IN && out_storage == OUT) return dispatch_vec<IN, MODE, OUT>(p, tp.vec, stream);
  DCQ_CASE(DCQ_F32, DCQ_F32)
  DCQ_CASE(DCQ_F16, DCQ_F16)
  DCQ_CASE(DCQ_BF16, DCQ_BF16)
  DCQ_CASE(DCQ_F32, DCQ_F16)
  DCQ_CASE(DCQ_F32, DCQ_BF16)
#undef DCQ_CASE
  set_error("unsupported storage combination in=%d out=%d", storage, out_storage);
  return DCQ_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// small element-wise kernels on scale arrays (a few thousand to a few hundred thousand values)
// ---------------------------------------------------------------------------------------------
struct ScaleParams {
  dcq_step step;
  dcq_rounding rounding;
  int64_t n, P, Q;
  const float* a;
  const float* b;
  float* c;
  float* d;
  const float* static_max;
  const float* static_min;
  int32_t flag;
  float value;
};

__global__ __launch_bounds__(kBlock) void range_to_scale_kernel(const ScaleParams p) {
  const dcq_step& st = p.step;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < p.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    float vmax, vmin = 0.0f;
    const bool zp = st.zero_domain != DCQ_ZERO_NONE;
    if (!p.flag && p.static_max) {  // static range: taken as is
      vmax = p.static_max[i];
      if (zp) vmin = p.static_min[i];
    } else {
      vmax = p.a[i];
      if (zp) vmin = p.d[i];
      if (p.b) {  // floating-point type with a zero point: symmetric about the group mean (data/range.py:334-341)
        const float vavg = p.b[i];
        float vmag = fmaxf(vmax - vavg, vavg - vmin);
        if (p.flag) vmag = vmag * p.value;
        vmax = vavg + vmag;
        vmin = vavg - vmag;
      } else if (p.flag) {
        vmax = vmax * p.value;
        if (zp) vmin = vmin * p.value;
      }
      if (st.zero_domain == DCQ_ZERO_PRE_SCALE) {
        vmax = fmaxf(vmax, 0.0f);
        vmin = fminf(vmin, 0.0f);
      }
      if (p.static_max) {
        vmax = fminf(vmax, p.static_max[i]);
        if (zp && p.static_min) vmin = fmaxf(vmin, p.static_min[i]);
      }
    }
    if (st.has_bound_max) vmax = fminf(vmax, st.bound_max);
    if (zp && st.has_bound_min) vmin = fmaxf(vmin, st.bound_min);
    const float span = zp ? vmax - vmin : vmax;
    p.c[i] = span / st.q_span;
    if (zp) p.d[i] = vmin;
  }
}

// one block per parent when reducing; grid-stride elementwise otherwise
__global__ __launch_bounds__(kBlock) void scale_level_kernel(const ScaleParams p) {
  __shared__ float cb[256];
  __shared__ float red[kBlock / kWave];
  if (p.rounding.kind == DCQ_ROUND_CODEBOOK) {
    for (int i = threadIdx.x; i < p.rounding.codebook_size; i += kBlock) cb[i] = p.rounding.codebook[i];
    __syncthreads();
  }
  float* s = p.c;
  if (!p.flag) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < p.n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
      p.d[i] = apply_rounding(p.rounding, fabsf(s[i]), cb);
    }
    return;
  }
  for (int64_t par = blockIdx.x; par < p.P; par += gridDim.x) {
    float* row = s + par * p.Q;
    float m = 0.0f;
    for (int64_t q = threadIdx.x; q < p.Q; q += kBlock) m = fmaxf(m, fabsf(row[q]));
    m = team_max(m, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) m = fmaxf(m, red[w]);
    const float level = apply_rounding(p.rounding, m / p.value, cb);
    if (threadIdx.x == 0) p.d[par] = level;
    for (int64_t q = threadIdx.x; q < p.Q; q += kBlock) row[q] = fabsf(row[q]) / level;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void round_values_kernel(const ScaleParams p) {
  __shared__ float cb[256];
  if (p.rounding.kind == DCQ_ROUND_CODEBOOK) {
    for (int i = threadIdx.x; i < p.rounding.codebook_size; i += kBlock) cb[i] = p.rounding.codebook[i];
    __syncthreads();
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < p.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    p.c[i] = apply_rounding(p.rounding, p.a[i], cb);
  }
}

__global__ __launch_bounds__(kBlock) void scale_join_kernel(const ScaleParams p) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < p.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    float v = p.a[i];
    if (p.b) v = v * p.b[i / p.Q];
    if (p.flag && v == 0.0f) v = 1.0f;
    p.c[i] = v;
  }
}

__global__ __launch_bounds__(kBlock) void zero_point_kernel(const ScaleParams p) {
  __shared__ float cb[256];
  const dcq_step& st = p.step;
  if (st.zero.kind == DCQ_ROUND_CODEBOOK) {
    for (int i = threadIdx.x; i < st.zero.codebook_size; i += kBlock) cb[i] = st.zero.codebook[i];
    __syncthreads();
  }
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < p.n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const float s = p.b[i], vmin = p.a[i];
    const float z = st.zero_domain == DCQ_ZERO_PRE_SCALE ? st.q_min - vmin / s : st.q_min * s - vmin;
    p.c[i] = apply_rounding(st.zero, z, cb);
  }
}

template <int ST>
__global__ __launch_bounds__(kBlock) void codebook_kernel(const typename Storage<ST>::elem_t* x,
                                                          typename Storage<ST>::elem_t* y, int64_t n,
                                                          const float* codebook, int C) {
  __shared__ float cb[256];
  for (int i = threadIdx.x; i < C; i += kBlock) cb[i] = codebook[i];
  __syncthreads();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    y[i] = Storage<ST>::store(codebook_nearest(Storage<ST>::load(x[i]), cb, C));
  }
}

// float64 tensors against a float64 table: the reference's kernel is instantiated for float and double
// (AT_DISPATCH_FLOATING_TYPES, csrc/quantize/quantize.cu:145) and compares in the tensor's own type
__global__ __launch_bounds__(kBlock) void codebook_f64_kernel(const double* x, double* y, int64_t n, const double* codebook,
                                                              int C) {
  __shared__ double cb[256];
  for (int i = threadIdx.x; i < C; i += kBlock) cb[i] = codebook[i];
  __syncthreads();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const double v = x[i];
    int lo = 0;
    double lval = cb[0];
#pragma unroll 1
    for (int bit = 1 << (31 - __clz(C)); bit; bit >>= 1) {
      const int next = lo | bit;
      const double nval = cb[next < C ? next : C - 1];
      const bool take = next < C && nval <= v;
      lo = take ? next : lo;
      lval = take ? nval : lval;
    }
    const double hval = cb[lo + (lo < C - 1)];
    y[i] = (v + v < lval + hval) ? lval : hval;
  }
}

template <int ST>
__global__ __launch_bounds__(kBlock) void prescale_kernel(const typename Storage<ST>::elem_t* x, int64_t n, int64_t S,
                                                          const float* outer, float lo, float hi, float* out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    out[i] = fminf(fmaxf(Storage<ST>::load(x[i]) / outer[i / S], lo), hi);
  }
}

// max |x| over a whole tensor into *out (which the caller has zeroed): magnitudes order like their bit patterns (NaN
// patterns above Inf above every finite value), so the reduction runs on integers -- for the 16-bit types on the packed
// words as loaded, one v_pk_max_u16 per two elements -- and the workgroups' maxima meet in one integer atomic; any order
// gives the same bits.  A pure read: four independent non-temporal 16-byte loads in flight per lane and two workgroups
// per compute unit (measured on 537 MB of bf16: 77 us = 7.0 TB/s; one plain load in flight on 2048 workgroups, the
// earlier form, 95 us; more or fewer workgroups lose 5-10 %).  A maximum at or above the Inf pattern means the tensor
// holds Inf or NaN: reported through `nonfinite` (|= 4), and the stored maximum is then Inf.
constexpr int kAbsmaxLoads = 4;

template <int ST>
__global__ __launch_bounds__(kBlock) void tensor_absmax_kernel(const typename Storage<ST>::elem_t* x0, int64_t n0,
                                                               int64_t head, float* out, int32_t* nonfinite) {
  constexpr int VEC = Storage<ST>::kVec;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __shared__ uint32_t red[kBlock / kWave];
  // the first `head` elements (up to the first 16-byte boundary) and the last n % VEC are read one by one
  const typename Storage<ST>::elem_t* x = x0 + head;
  const int64_t n = n0 - head;
  const u32x4* x4 = reinterpret_cast<const u32x4*>(x);
  const int64_t chunks = n / VEC;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  uint32_t acc = 0;  // float32: magnitude bits; 16-bit types: one running maximum per half word
  for (int64_t c = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; c < chunks; c += stride * kAbsmaxLoads) {
    u32x4 r[kAbsmaxLoads];
#pragma unroll
    for (int u = 0; u < kAbsmaxLoads; ++u) {
      const int64_t i = c + u * stride;
      r[u] = i < chunks ? __builtin_nontemporal_load(x4 + i) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < kAbsmaxLoads; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ST == DCQ_F32) {
          acc = max(acc, r[u][k] & 0x7fffffffu);
        } else {
          const uint32_t a = r[u][k] & 0x7fff7fffu;
          uint32_t m;
          asm("v_pk_max_u16 %0, %1, %2" : "=v"(m) : "v"(acc), "v"(a));
          acc = m;
        }
      }
    }
  }
  // magnitude bits widened to the float32 pattern of the same value (bf16: shift; f16: convert, Inf / NaN kept)
  auto widen = [](uint32_t half) -> uint32_t {
    if (ST == DCQ_BF16) return half << 16;
    return __float_as_uint(f16_to_f32(static_cast<uint16_t>(half)));
  };
  uint32_t top = ST == DCQ_F32 ? acc : max(widen(acc & 0xffffu), widen(acc >> 16));
  if (ST != DCQ_F32) top &= 0x7fffffffu;
  if (blockIdx.x == 0) {
    for (int64_t i = chunks * VEC + threadIdx.x; i < n; i += kBlock)
      top = max(top, __float_as_uint(fabsf(Storage<ST>::load(x[i]))));
    if (threadIdx.x < head) top = max(top, __float_as_uint(fabsf(Storage<ST>::load(x0[threadIdx.x]))));
  }
  for (int off = kWave / 2; off > 0; off >>= 1) top = max(top, static_cast<uint32_t>(__shfl_xor(static_cast<int>(top), off, kWave)));
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = top;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) top = max(top, red[w]);
    if (top >= 0x7f800000u) {
      if (nonfinite != nullptr) atomicOr(nonfinite, 4);
      top = 0x7f800000u;
    }
    atomicMax(reinterpret_cast<unsigned int*>(out), top);
  }
}

static unsigned small_grid(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return static_cast<unsigned>(b);
}

// flags |= 1 when any x[i] is NaN, |= 2 when any is +-Inf; one read of x (float32), 16-byte loads, one atomic per wave
// that saw something
__global__ __launch_bounds__(256) void nonfinite_flags_kernel(const float* x, int64_t n, int32_t* flags) {
  const int64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  uint32_t seen = 0;
  auto look = [&](float v) {
    const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
    seen |= b > 0x7f800000u ? 1u : (b == 0x7f800000u ? 2u : 0u);
  };
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const float4 v = x4[i];
    look(v.x), look(v.y), look(v.z), look(v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) look(x[n4 * 4 + threadIdx.x]);
  if (seen) atomicOr(flags, static_cast<int32_t>(seen));
}

}  // namespace dcq

// =============================================================================================
// C ABI
// =============================================================================================
using namespace dcq;

extern "C" {

const char* dcq_version(void) { return "dcq 0.1 (gfx950)"; }
const char* dcq_last_error(void) { return g_error; }

int dcq_round_to_codebook(const void* x, void* y, int64_t n, int32_t storage, const void* codebook_values,
                          int32_t codebook_size, dcq_stream_t stream) {
  const float* codebook = static_cast<const float*>(codebook_values);
  DCQ_REQUIRE(n >= 0, DCQ_ERR_ARGUMENT, "n < 0");
  if (n == 0) return DCQ_OK;
  DCQ_REQUIRE(x && y && codebook, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(codebook_size > 0 && codebook_size <= 256, DCQ_ERR_ARGUMENT, "codebook size %d not in 1..256",
              codebook_size);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = small_grid(n);
  switch (storage) {
    case DCQ_F32:
      hipLaunchKernelGGL(codebook_kernel<DCQ_F32>, dim3(grid), dim3(kBlock), 0, s, static_cast<const float*>(x),
                         static_cast<float*>(y), n, codebook, codebook_size);
      break;
    case DCQ_F16:
      hipLaunchKernelGGL(codebook_kernel<DCQ_F16>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t*>(x),
                         static_cast<uint16_t*>(y), n, codebook, codebook_size);
      break;
    case DCQ_BF16:
      hipLaunchKernelGGL(codebook_kernel<DCQ_BF16>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t*>(x),
                         static_cast<uint16_t*>(y), n, codebook, codebook_size);
      break;
    case DCQ_F64:
      hipLaunchKernelGGL(codebook_f64_kernel, dim3(grid), dim3(kBlock), 0, s, static_cast<const double*>(x),
                         static_cast<double*>(y), n, static_cast<const double*>(codebook_values), codebook_size);
      break;
    default:
      set_error("bad storage %d", storage);
      return DCQ_ERR_ARGUMENT;
  }
  DCQ_LAUNCH_CHECK("codebook_kernel launch");
  return DCQ_OK;
}

int dcq_fake_quant_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                          const float* range_max, const float* range_min, int32_t has_ratio, float ratio,
                          void* dequant, float* codes, float* scale, float* zero, int32_t* nonfinite,
                          dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(x != nullptr, DCQ_ERR_ARGUMENT, "x is NULL");
  DCQ_REQUIRE(storage >= DCQ_F32 && storage <= DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage %d", storage);
  DCQ_REQUIRE(range_max || !needs_mean_centred_range(step), DCQ_ERR_UNSUPPORTED,
              "floating-point dtype with a zero point: the fused kernel needs the (mean-centred) range as input");
  DCQ_REQUIRE(!(range_max && step->zero_domain != DCQ_ZERO_NONE && !range_min), DCQ_ERR_ARGUMENT,
              "a static range with a zero point needs range_min");
  DCQ_REQUIRE(!(range_max && has_ratio), DCQ_ERR_UNSUPPORTED,
              "ratio together with a bounding range: use dcq_group_range + dcq_range_to_scale");
  rc = try_launch_fused_int(*step, x, storage, G, S, range_max, range_min, has_ratio, ratio, dequant, codes, scale,
                            zero, static_cast<hipStream_t>(stream), nullptr, nonfinite);
  if (rc != DCQ_ERR_UNSUPPORTED) return rc;
  if (range_max == nullptr) {
    rc = try_launch_fused_table(*step, x, storage, G, S, has_ratio, ratio, nullptr, nullptr, 0.0f, dequant, codes,
                                nullptr, nullptr, scale, zero, static_cast<hipStream_t>(stream), nonfinite);
    if (rc != DCQ_ERR_UNSUPPORTED) return rc;
  }
  GroupParams p{};
  p.step = *step;
  p.x = x;
  p.G = G;
  p.S = S;
  p.range_max = range_max;
  p.range_min = range_min;
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.dequant = dequant;
  p.codes = codes;
  p.scale_out = scale;
  p.zero_out = zero;
  p.nonfinite = nonfinite;
  return dispatch_storage<kFused>(p, storage, storage, static_cast<hipStream_t>(stream));
}

int dcq_fake_quant_groups_outer(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                                const float* outer, float outer_lo, float outer_hi, int32_t deq_clamp, float deq_lo,
                                float deq_hi, const float* range_max, const float* range_min, int32_t has_ratio,
                                float ratio, void* dequant, int32_t out_storage, float* codes, float* scale, float* zero,
                                int32_t* nonfinite, dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(x && outer, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(storage >= DCQ_F32 && storage <= DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage %d", storage);
  DCQ_REQUIRE(!needs_mean_centred_range(step), DCQ_ERR_UNSUPPORTED,
              "floating-point dtype with a zero point after a progressive step");
  DCQ_REQUIRE(!(range_max && step->zero_domain != DCQ_ZERO_NONE && !range_min), DCQ_ERR_ARGUMENT,
              "a static range with a zero point needs range_min");
  DCQ_REQUIRE(!(range_max && has_ratio), DCQ_ERR_UNSUPPORTED, "ratio together with a static range");
  if (dequant == nullptr || out_storage == storage) {
    const OuterStep prev{outer, outer_lo, outer_hi, deq_clamp, deq_lo, deq_hi};
    rc = try_launch_fused_int(*step, x, storage, G, S, range_max, range_min, has_ratio, ratio, dequant, codes, scale,
                              zero, static_cast<hipStream_t>(stream), &prev, nonfinite);
    if (rc != DCQ_ERR_UNSUPPORTED) return rc;
  }
  GroupParams p{};
  p.step = *step;
  p.x = x;
  p.G = G;
  p.S = S;
  p.range_max = range_max;
  p.range_min = range_min;
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.outer = outer;
  p.outer_lo = outer_lo;
  p.outer_hi = outer_hi;
  p.deq_clamp = deq_clamp;
  p.deq_lo = deq_lo;
  p.deq_hi = deq_hi;
  p.dequant = dequant;
  p.codes = codes;
  p.scale_out = scale;
  p.zero_out = zero;
  p.nonfinite = nonfinite;
  return dispatch_storage<kFused>(p, storage, dequant ? out_storage : storage, static_cast<hipStream_t>(stream));
}

int dcq_fake_quant_groups_two_level(const dcq_step* step, const dcq_rounding* level0, float level0_span, const void* x,
                                    int32_t storage, int64_t G, int64_t S, int32_t has_ratio, float ratio,
                                    float* tensor_absmax, void* dequant, float* codes, float* level0_out,
                                    float* level1_out, float* scale, int32_t* nonfinite, dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(level0 != nullptr && level0_span > 0.0f, DCQ_ERR_ARGUMENT, "first scale level is missing");
  DCQ_REQUIRE(level0->kind == DCQ_ROUND_F32 || level0->kind == DCQ_ROUND_F16 || level0->kind == DCQ_ROUND_BF16,
              DCQ_ERR_UNSUPPORTED, "the per-tensor scale level must be a float type");
  DCQ_REQUIRE(step->zero_domain == DCQ_ZERO_NONE, DCQ_ERR_UNSUPPORTED, "two scale levels with a zero point");
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(x && tensor_absmax, DCQ_ERR_ARGUMENT, "NULL buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = hip_check(hipMemsetAsync(tensor_absmax, 0, 4, s), "hipMemsetAsync(tensor_absmax)");
  if (rc != DCQ_OK) return rc;
  const int64_t n = G * S;
  const uintptr_t width = storage == DCQ_F32 ? 4 : 2;
  DCQ_REQUIRE(reinterpret_cast<uintptr_t>(x) % width == 0, DCQ_ERR_ARGUMENT, "x is not aligned to its element type");
  int64_t head = static_cast<int64_t>(((16 - (reinterpret_cast<uintptr_t>(x) & 15u)) & 15u) / width);
  if (head > n) head = n;
  // two workgroups per compute unit, each lane with kAbsmaxLoads loads in flight (see the kernel)
  const int64_t lane_loads = (n / (storage == DCQ_F32 ? 4 : 8) + kAbsmaxLoads - 1) / kAbsmaxLoads;
  int64_t wanted = (lane_loads + kBlock - 1) / kBlock;
  wanted = wanted < 1 ? 1 : (wanted > 512 ? 512 : wanted);
  const unsigned grid = static_cast<unsigned>(wanted);
  switch (storage) {
    case DCQ_F32:
      hipLaunchKernelGGL(tensor_absmax_kernel<DCQ_F32>, dim3(grid), dim3(kBlock), 0, s, static_cast<const float*>(x), n,
                         head, tensor_absmax, nonfinite);
      break;
    case DCQ_F16:
      hipLaunchKernelGGL(tensor_absmax_kernel<DCQ_F16>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t*>(x), n,
                         head, tensor_absmax, nonfinite);
      break;
    case DCQ_BF16:
      hipLaunchKernelGGL(tensor_absmax_kernel<DCQ_BF16>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t*>(x),
                         n, head, tensor_absmax, nonfinite);
      break;
    default:
      set_error("bad storage %d", storage);
      return DCQ_ERR_ARGUMENT;
  }
  DCQ_LAUNCH_CHECK("tensor_absmax_kernel launch");
  rc = try_launch_fused_table(*step, x, storage, G, S, has_ratio, ratio, tensor_absmax, level0, level0_span, dequant,
                              codes, level0_out, level1_out, scale, nullptr, s, nullptr);
  if (rc != DCQ_ERR_UNSUPPORTED) return rc;
  GroupParams p{};
  p.step = *step;
  p.x = x;
  p.G = G;
  p.S = S;
  p.has_ratio = has_ratio;
  p.ratio = ratio;
  p.tensor_absmax = tensor_absmax;
  p.level0 = *level0;
  p.level0_span = level0_span;
  p.level0_out = level0_out;
  p.level1_out = level1_out;
  p.dequant = dequant;
  p.codes = codes;
  p.scale_out = scale;
  return dispatch_storage<kFused>(p, storage, storage, s);
}

int dcq_group_range(const void* x, int32_t storage, int64_t G, int64_t S, int32_t zero_domain, const float* outer,
                    float outer_lo, float outer_hi, float* vmax, float* vmin, int32_t* nonfinite, dcq_stream_t stream) {
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(x && vmax, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(storage >= DCQ_F32 && storage <= DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage %d", storage);
  DCQ_REQUIRE(zero_domain == DCQ_ZERO_NONE || vmin, DCQ_ERR_ARGUMENT, "vmin is NULL");
  GroupParams p{};
  p.step.zero_domain = zero_domain;
  p.step.tensor.kind = DCQ_ROUND_INT;
  p.x = x;
  p.G = G;
  p.S = S;
  p.outer = outer;
  p.outer_lo = outer_lo;
  p.outer_hi = outer_hi;
  p.vmax_out = vmax;
  p.vmin_out = vmin;
  p.nonfinite = nonfinite;
  return dispatch_storage<kRangeOnly>(p, storage, storage, static_cast<hipStream_t>(stream));
}

int dcq_range_to_scale(const dcq_step* step, int64_t n, const float* vmax, float* vmin, const float* vavg,
                       int32_t has_ratio, float ratio, const float* static_max, const float* static_min,
                       float* linear_scale, dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  if (n == 0) return DCQ_OK;
  const bool zp = step->zero_domain != DCQ_ZERO_NONE;
  DCQ_REQUIRE(n > 0 && linear_scale && (vmax || static_max), DCQ_ERR_ARGUMENT, "bad arguments");
  DCQ_REQUIRE(!zp || vmin, DCQ_ERR_ARGUMENT, "vmin is NULL (it also receives the final range minimum)");
  DCQ_REQUIRE(!(static_max && zp && !has_ratio && !static_min), DCQ_ERR_ARGUMENT, "static range needs static_min");
  DCQ_REQUIRE(!vavg || (zp && vmax), DCQ_ERR_ARGUMENT, "a group mean goes with a measured range and a zero point");
  ScaleParams p{};
  p.step = *step;
  p.n = n;
  p.a = vmax;
  p.b = vavg;
  p.d = vmin;
  p.c = linear_scale;
  p.static_max = static_max;
  p.static_min = static_min;
  p.flag = has_ratio;
  p.value = ratio;
  hipLaunchKernelGGL(range_to_scale_kernel, dim3(small_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  DCQ_LAUNCH_CHECK("range_to_scale_kernel launch");
  return DCQ_OK;
}

int dcq_scale_level(const dcq_rounding* rounding, float* s, int64_t P, int64_t Q, int32_t reduce, float span,
                    float* level, dcq_stream_t stream) {
  DCQ_REQUIRE(rounding && s && level && P >= 0 && Q >= 1, DCQ_ERR_ARGUMENT, "bad arguments");
  if (P == 0) return DCQ_OK;
  ScaleParams p{};
  p.rounding = *rounding;
  p.P = P;
  p.Q = Q;
  p.n = P * Q;
  p.c = s;
  p.d = level;
  p.flag = reduce;
  p.value = span;
  DCQ_REQUIRE(reduce || Q == 1, DCQ_ERR_ARGUMENT, "reduce == 0 requires Q == 1");
  const unsigned grid = reduce ? static_cast<unsigned>(P < 4096 ? P : 4096) : small_grid(p.n);
  hipLaunchKernelGGL(scale_level_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  DCQ_LAUNCH_CHECK("scale_level_kernel launch");
  return DCQ_OK;
}

int dcq_round_values(const dcq_rounding* rounding, const float* in, float* out, int64_t n, dcq_stream_t stream) {
  DCQ_REQUIRE(rounding && n >= 0, DCQ_ERR_ARGUMENT, "bad arguments");
  if (n == 0) return DCQ_OK;
  DCQ_REQUIRE(in && out, DCQ_ERR_ARGUMENT, "NULL buffer");
  ScaleParams p{};
  p.rounding = *rounding;
  p.n = n;
  p.a = in;
  p.c = out;
  hipLaunchKernelGGL(round_values_kernel, dim3(small_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  DCQ_LAUNCH_CHECK("round_values_kernel launch");
  return DCQ_OK;
}

int dcq_scale_join(const float* a, const float* b, int64_t n, int64_t Q, int32_t fix_zero, float* compound,
                   dcq_stream_t stream) {
  DCQ_REQUIRE(a && compound && n >= 0 && Q >= 1, DCQ_ERR_ARGUMENT, "bad arguments");
  if (n == 0) return DCQ_OK;
  ScaleParams p{};
  p.a = a;
  p.b = b;
  p.c = compound;
  p.n = n;
  p.Q = Q;
  p.flag = fix_zero;
  hipLaunchKernelGGL(scale_join_kernel, dim3(small_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  DCQ_LAUNCH_CHECK("scale_join_kernel launch");
  return DCQ_OK;
}

int dcq_zero_point(const dcq_step* step, int64_t n, const float* vmin, const float* scale, float* zero,
                   dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(step->zero_domain != DCQ_ZERO_NONE, DCQ_ERR_ARGUMENT, "no zero point in this step");
  DCQ_REQUIRE(n >= 0 && vmin && scale && zero, DCQ_ERR_ARGUMENT, "bad arguments");
  if (n == 0) return DCQ_OK;
  ScaleParams p{};
  p.step = *step;
  p.n = n;
  p.a = vmin;
  p.b = scale;
  p.c = zero;
  hipLaunchKernelGGL(zero_point_kernel, dim3(small_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
  DCQ_LAUNCH_CHECK("zero_point_kernel launch");
  return DCQ_OK;
}

int dcq_quantize_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S, const float* scale,
                        const float* zero, const float* outer, float outer_lo, float outer_hi, int32_t deq_clamp,
                        float deq_lo, float deq_hi, void* dequant, int32_t out_storage, float* codes,
                        const float* round_delta, int32_t* nonfinite, dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(x && scale, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(step->zero_domain == DCQ_ZERO_NONE || zero, DCQ_ERR_ARGUMENT, "zero is NULL");
  GroupParams p{};
  p.step = *step;
  p.x = x;
  p.G = G;
  p.S = S;
  p.scale_in = scale;
  p.zero_in = zero;
  p.outer = outer;
  p.outer_lo = outer_lo;
  p.outer_hi = outer_hi;
  p.deq_clamp = deq_clamp;
  p.deq_lo = deq_lo;
  p.deq_hi = deq_hi;
  p.dequant = dequant;
  p.codes = codes;
  DCQ_REQUIRE(round_delta == nullptr || step->tensor.kind == DCQ_ROUND_INT, DCQ_ERR_UNSUPPORTED,
              "round_delta is defined for integer types only");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(round_delta) & 15u) == 0, DCQ_ERR_ARGUMENT, "round_delta must be 16-byte aligned");
  p.round_delta = round_delta;
  p.nonfinite = nonfinite;
  return dispatch_storage<kQuantOnly>(p, storage, dequant ? out_storage : storage, static_cast<hipStream_t>(stream));
}

int dcq_prescale_groups(const void* x, int32_t storage, int64_t G, int64_t S, const float* outer, float lo, float hi,
                        float* out, dcq_stream_t stream) {
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(x && outer && out, DCQ_ERR_ARGUMENT, "NULL buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n = G * S;
  const unsigned grid = small_grid(n);
  switch (storage) {
    case DCQ_F32:
      hipLaunchKernelGGL(prescale_kernel<DCQ_F32>, dim3(grid), dim3(kBlock), 0, s, static_cast<const float*>(x), n, S,
                         outer, lo, hi, out);
      break;
    case DCQ_F16:
      hipLaunchKernelGGL(prescale_kernel<DCQ_F16>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t*>(x), n,
                         S, outer, lo, hi, out);
      break;
    case DCQ_BF16:
      hipLaunchKernelGGL(prescale_kernel<DCQ_BF16>, dim3(grid), dim3(kBlock), 0, s, static_cast<const uint16_t*>(x), n,
                         S, outer, lo, hi, out);
      break;
    default:
      set_error("bad storage %d", storage);
      return DCQ_ERR_ARGUMENT;
  }
  DCQ_LAUNCH_CHECK("prescale_kernel launch");
  return DCQ_OK;
}

int dcq_dequantize_groups(const dcq_step* step, const float* codes, int64_t G, int64_t S, const float* scale,
                          const float* zero, const float* outer, int32_t deq_clamp, float deq_lo, float deq_hi,
                          void* dequant, int32_t out_storage, dcq_stream_t stream) {
  int rc = validate_step(step);
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(G >= 0 && S >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (G == 0 || S == 0) return DCQ_OK;
  DCQ_REQUIRE(codes && scale && dequant, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(step->zero_domain == DCQ_ZERO_NONE || zero, DCQ_ERR_ARGUMENT, "zero is NULL");
  GroupParams p{};
  p.step = *step;
  p.x = codes;
  p.G = G;
  p.S = S;
  p.scale_in = scale;
  p.zero_in = zero;
  p.outer = outer;
  p.deq_clamp = deq_clamp;
  p.deq_lo = deq_lo;
  p.deq_hi = deq_hi;
  p.dequant = dequant;
  return dispatch_storage<kDequantOnly>(p, DCQ_F32, out_storage, static_cast<hipStream_t>(stream));
}

int dcq_nonfinite_flags(const float* x, int64_t n, int32_t* flags, dcq_stream_t stream) {
  DCQ_REQUIRE(n >= 0 && flags != nullptr, DCQ_ERR_ARGUMENT, "bad arguments");
  if (n == 0) return DCQ_OK;
  DCQ_REQUIRE(x != nullptr && (reinterpret_cast<uintptr_t>(x) & 15u) == 0, DCQ_ERR_ARGUMENT,
              "x must be 16-byte aligned");
  int64_t blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(nonfinite_flags_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, n, flags);
  DCQ_LAUNCH_CHECK("nonfinite_flags_kernel launch");
  return DCQ_OK;
}

}  // extern "C"
