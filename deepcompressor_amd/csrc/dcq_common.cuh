// Device-side helpers shared by the quantization kernels (gfx950 only).
//
// Numerical contract: everything here is plain IEEE float32 with round-to-nearest-even and must be
// compiled with -ffp-contract=off (a fused multiply-add would change results that the reference
// computes as two rounded operations).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>

#include "dcq.h"

namespace dcq {

constexpr int kWave = 64;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---------------------------------------------------------------------------------------------
// host-side error reporting
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_check(hipError_t e, const char* what);

// The previous step of a progressive quantizer as the last step sees it: the step quantizes
// clamp(x / scale[g], lo, hi) and multiplies its (optionally clamped) de-quantized values by scale[g].
struct OuterStep {
  const float* scale;  // [G], device
  float lo, hi;
  int deq_clamp;
  float deq_lo, deq_hi;
};

// lean fused kernel (dcq_fused.hip); DCQ_ERR_UNSUPPORTED means "use the general kernel"
int try_launch_fused_int(const dcq_step& st, const void* x, int storage, int64_t G, int64_t S, const float* range_max,
                         const float* range_min, int has_ratio, float ratio, void* dequant, float* codes, float* scale,
                         float* zero, hipStream_t stream, const OuterStep* outer = nullptr, int32_t* nonfinite = nullptr);

// lean kernel for value-table (floating-point) dtypes without zero point, one or two scale levels (dcq_fused.hip)
// float32 GEMM (dcq_gemm_f32) with the arithmetic path chosen by the caller: the float32 matrix instructions, or
// three-term bf16 splits on the bf16 matrix pipe (same accuracy, other roundings); kGemmAuto picks by problem size.
// `scratch`: NULL, or two_term_scratch_bytes(M, N, K) bytes (256-byte aligned), with which a product on the split path
// uses two float16 terms per operand (scaled per row of A and column of B) instead of three bfloat16 ones.
enum { kGemmAuto = 0, kGemmFloat32 = 1, kGemmSplit = 2 };
int gemm_f32_on_path(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, float alpha, float beta, hipStream_t stream, int path,
                     unsigned char* scratch = nullptr);
size_t two_term_scratch_bytes(int64_t M, int64_t N, int64_t K);

int try_launch_fused_table(const dcq_step& st, const void* x, int storage, int64_t G, int64_t S, int has_ratio,
                           float ratio, const float* tensor_absmax, const dcq_rounding* level0, float level0_span,
                           void* dequant, float* codes, float* level0_out, float* level1_out, float* scale,
                           float* zero, hipStream_t stream, int32_t* nonfinite = nullptr);

// float -> storage dtype -> double of a power's exponent: torch rounds the scalar exponent of Tensor.pow_ to the tensor's
// dtype (dcq_search.hip)
double round_degree_to_storage(double degree, int storage);

#define DCQ_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      ::dcq::set_error(__VA_ARGS__); \
      return (code);                 \
    }                                \
  } while (0)

#define DCQ_LAUNCH_CHECK(name)                                       \
  do {                                                               \
    int _rc = ::dcq::hip_check(hipGetLastError(), name);             \
    if (_rc != DCQ_OK) return _rc;                                   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// storage types
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(static_cast<uint32_t>(b) << 16); }

__device__ __forceinline__ uint16_t f32_to_bf16(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);  // keep NaN a NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

__device__ __forceinline__ float f16_to_f32(uint16_t b) { return __half2float(__ushort_as_half(b)); }
__device__ __forceinline__ uint16_t f32_to_f16(float x) { return __half_as_ushort(__float2half_rn(x)); }

template <int ST>
struct Storage;

template <>
struct Storage<DCQ_F32> {
  using elem_t = float;
  static constexpr int kVec = 4;  // elements per 16-byte access
  static __device__ __forceinline__ float load(elem_t v) { return v; }
  static __device__ __forceinline__ elem_t store(float v) { return v; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
template <>
struct Storage<DCQ_F16> {
  using elem_t = uint16_t;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float load(elem_t v) { return f16_to_f32(v); }
  static __device__ __forceinline__ elem_t store(float v) { return f32_to_f16(v); }
  static __device__ __forceinline__ float round(float v) { return f16_to_f32(f32_to_f16(v)); }
};
template <>
struct Storage<DCQ_BF16> {
  using elem_t = uint16_t;
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float load(elem_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ elem_t store(float v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// A 16-byte (or, for VEC == 1, single element) chunk of a tensor.
template <typename E, int VEC>
struct alignas(sizeof(E) * VEC) Chunk {
  E v[VEC];
};

template <typename E, int VEC>
__device__ __forceinline__ Chunk<E, VEC> load_chunk(const E* p) {
  return *reinterpret_cast<const Chunk<E, VEC>*>(p);
}
template <typename E, int VEC>
__device__ __forceinline__ void store_chunk(E* p, const Chunk<E, VEC>& c) {
  *reinterpret_cast<Chunk<E, VEC>*>(p) = c;
}

// ---------------------------------------------------------------------------------------------
// codebook search: largest index lo with cb[lo] <= x (0 if none); tie between the two neighbours goes up
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float codebook_nearest(float x, const float* cb, int C) {
  int lo = 0;
  float lval = cb[0];
#pragma unroll 1
  for (int bit = 1 << (31 - __clz(C)); bit; bit >>= 1) {
    const int next = lo | bit;
    const float nval = cb[next < C ? next : C - 1];
    const bool take = next < C && nval <= x;
    lo = take ? next : lo;
    lval = take ? nval : lval;
  }
  const float hval = cb[lo + (lo < C - 1)];
  return (x + x < lval + hval) ? lval : hval;
}

// One application of a dcq_rounding.  `cb` points at the (LDS or global) copy of r.codebook.
__device__ __forceinline__ float apply_rounding(const dcq_rounding& r, float v, const float* cb) {
  float y;
  switch (r.kind) {
    case DCQ_ROUND_F16:
      y = f16_to_f32(f32_to_f16(v));
      break;
    case DCQ_ROUND_BF16:
      y = bf16_to_f32(f32_to_bf16(v));
      break;
    case DCQ_ROUND_INT:
      return fminf(fmaxf(rintf(v), r.lo), r.hi);
    case DCQ_ROUND_CODEBOOK:
      y = codebook_nearest(fminf(fmaxf(v, r.lo), r.hi), cb, r.codebook_size);
      break;
    case DCQ_ROUND_EXP:
      return exp2f(fminf(fmaxf(floorf(log2f(v)), r.lo), r.hi));
    default:
      y = v;
      break;
  }
  if (r.has_clamp) y = fminf(fmaxf(y, r.clamp_lo), r.clamp_hi);
  return y;
}

// Integer fast path of apply_rounding (kind known at compile time).
__device__ __forceinline__ float round_int(float v, float lo, float hi) { return fminf(fmaxf(rintf(v), lo), hi); }

// ---------------------------------------------------------------------------------------------
// per-group scale / zero from a (already ratio-scaled) range, single scale level
// ---------------------------------------------------------------------------------------------
struct ScaleZero {
  float s, z;
};

__device__ __forceinline__ ScaleZero scale_zero_from_range(const dcq_step& st, float vmax, float vmin,
                                                           const float* s_cb, const float* z_cb) {
  if (st.has_bound_max) vmax = fminf(vmax, st.bound_max);
  if (st.zero_domain != DCQ_ZERO_NONE && st.has_bound_min) vmin = fmaxf(vmin, st.bound_min);
  const float span = st.zero_domain != DCQ_ZERO_NONE ? vmax - vmin : vmax;
  float s = apply_rounding(st.scale, fabsf(span / st.q_span), s_cb);
  if (s == 0.0f) s = 1.0f;
  float z = 0.0f;
  if (st.zero_domain == DCQ_ZERO_PRE_SCALE) {
    z = apply_rounding(st.zero, st.q_min - vmin / s, z_cb);
  } else if (st.zero_domain == DCQ_ZERO_POST_SCALE) {
    z = apply_rounding(st.zero, st.q_min * s - vmin, z_cb);
  }
  return {s, z};
}

template <int ZD>
__device__ __forceinline__ float to_code_input(float x, float s, float z) {
  float t = x;
  if (ZD == DCQ_ZERO_POST_SCALE) t = t + z;
  t = t / s;
  if (ZD == DCQ_ZERO_PRE_SCALE) t = t + z;
  return t;
}

template <int ZD>
__device__ __forceinline__ float from_code(float q, float s, float z) {
  if (ZD == DCQ_ZERO_PRE_SCALE) return (q - z) * s;
  if (ZD == DCQ_ZERO_POST_SCALE) return q * s - z;
  return q * s;
}

// ---------------------------------------------------------------------------------------------
// IEEE-exact x / s for many x and one s.
//
// The hardware's own correctly-rounded float32 division is: y = refined reciprocal of s (rcp + one Newton
// step), q0 = x*y, two residual corrections q <- q + (x - s*q)*y, wrapped in a pre-scaling / fix-up pair that
// only matters when an intermediate could overflow or go subnormal.  The reciprocal part depends on s alone,
// so it is hoisted out of the element loop, and the wrapping is dropped under a guard that makes it
// unnecessary: 2^-60 <= s <= 2^60 and |x| <= 2^60.  (Then every intermediate of the corrections is a normal
// number unless |x| < 2^-96, in which case |x / s| < 2^-36 and its last bit cannot change round(x / s [+ z]).)
// Groups outside the guard take the plain division.  dcq_selftest_division checks the equivalence on the GPU.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Divisor {
  float s, y;
  bool fast;
};

__device__ __forceinline__ Divisor make_divisor(float s, float abs_bound) {
  Divisor d;
  d.s = s;
  const float y0 = __builtin_amdgcn_rcpf(s);
  const float e = __builtin_fmaf(-s, y0, 1.0f);
  d.y = __builtin_fmaf(e, y0, y0);
  d.fast = s >= 0x1p-60f && s <= 0x1p60f && abs_bound <= 0x1p60f;
  return d;
}

__device__ __forceinline__ f32x2 div_fast2(f32x2 x, const Divisor& d) {
  const f32x2 y = {d.y, d.y}, ns = {-d.s, -d.s};
  f32x2 q = x * y;
  f32x2 r = __builtin_elementwise_fma(ns, q, x);
  q = __builtin_elementwise_fma(r, y, q);
  r = __builtin_elementwise_fma(ns, q, x);
  return __builtin_elementwise_fma(r, y, q);
}

// The same quotient with ONE correction step, for a divisor whose significand has at most 16 bits (a scale rounded
// to float16 / bfloat16, an integer scale of up to 8 bits, a product of a float8 and a 16-bit scale) and any float32
// x, inside the same guard.  Why one step is enough then: with y within one ulp of 1 / s, q = RN(x y) is within
// 2^-22.4 |x / s|; r = x - s q is a multiple of ulp16(s) ulp(q) below 2^18 such units, so the fused multiply-add
// delivers it exactly; q + r y differs from the exact quotient by |r / s| 2^-23 < 2^-45 |x / s|; and the exact
// quotient of a 24-bit by a 16-bit significand stays at least 2^-41 |x / s| away from every midpoint between two
// float32 values (x - s m is a non-zero multiple of ulp16(s) ulp25(m) for a 25-bit midpoint m: s m has an odd
// significand of more than 24 bits and cannot equal x), so rounding q + r y rounds the exact quotient.
__device__ __forceinline__ f32x2 div_short2(f32x2 x, const Divisor& d) {
  const f32x2 y = {d.y, d.y}, ns = {-d.s, -d.s};
  const f32x2 q = x * y;
  const f32x2 r = __builtin_elementwise_fma(ns, q, x);
  return __builtin_elementwise_fma(r, y, q);
}

__device__ __forceinline__ float div_short(float x, const Divisor& d) {
  const float q = x * d.y;
  const float r = __builtin_fmaf(-d.s, q, x);
  return __builtin_fmaf(r, d.y, q);
}

// whether a scale rounded by `kind` (a dcq_rounding kind) has a significand of at most 16 bits
__host__ __device__ constexpr bool short_significand(int kind) {
  return kind == DCQ_ROUND_F16 || kind == DCQ_ROUND_BF16 || kind == DCQ_ROUND_INT;
}

__device__ __forceinline__ float div_fast(float x, const Divisor& d) {
  float q = x * d.y;
  float r = __builtin_fmaf(-d.s, q, x);
  q = __builtin_fmaf(r, d.y, q);
  r = __builtin_fmaf(-d.s, q, x);
  return __builtin_fmaf(r, d.y, q);
}

// xor-butterfly reductions over `width` adjacent lanes (width a power of two <= 64)
__device__ __forceinline__ float team_max(float v, int width) {
  for (int off = width >> 1; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ float team_min(float v, int width) {
  for (int off = width >> 1; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}

}  // namespace dcq
