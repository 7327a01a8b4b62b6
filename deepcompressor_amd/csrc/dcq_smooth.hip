// Per-channel statistics and per-channel scaling for smooth quantization (gfx950).
//   channel_metric_kernel : ChannelMetric._abs_max / _abs_sum / _square_sum  (deepcompressor/calib/metric.py:31-86)
//   scale_channels_kernel : smooth_upscale_param / smooth_downscale_param / ActivationSmoother.process
//                           (deepcompressor/calib/smooth.py:56-78, 852-891)
// Both are single passes over the tensor: HBM-bound.
#include "dcq_common.cuh"

namespace dcq {

enum { kMetricAbsMax = 0, kMetricAbsSum = 1, kMetricSquareSum = 2 };

struct MetricParams {
  const void* x;  // [N][C][inner]
  int64_t N, C, inner;
  int32_t mode;
  float* out_max;    // [C] (abs max: integer atomic max on the non-negative float bits; caller zeroes it)
  double* out_sum;   // [C] (sums: float32 partial per workgroup, added in float64; caller zeroes it)
  int64_t rows_per_block;
};

__device__ __forceinline__ float metric_term(float v, int mode) {
  if (mode == kMetricSquareSum) return v * v;
  return fabsf(v);
}

// inner == 1: x is [N][C] row-major.  One thread owns VEC consecutive channels and walks a slab of rows.
template <int ST, int VEC>
__global__ __launch_bounds__(256) void channel_metric_rows_kernel(const MetricParams p) {
  using S = Storage<ST>;
  using E = typename S::elem_t;
  const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * VEC;
  if (c0 >= p.C) return;
  const int64_t r_begin = static_cast<int64_t>(blockIdx.y) * p.rows_per_block;
  const int64_t r_end = min(p.N, r_begin + p.rows_per_block);
  const E* base = static_cast<const E*>(p.x);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
  for (int64_t r = r_begin; r < r_end; ++r) {
    const Chunk<E, VEC> ch = load_chunk<E, VEC>(base + r * p.C + c0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float t = metric_term(S::load(ch.v[i]), p.mode);
      // fmaxf would drop a NaN; the reference's amax propagates it
      acc[i] = p.mode == kMetricAbsMax ? ((t > acc[i] || t != t) ? t : acc[i]) : acc[i] + t;
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    if (c0 + i >= p.C) break;
    if (p.mode == kMetricAbsMax) {
      atomicMax(reinterpret_cast<int*>(p.out_max) + c0 + i, __float_as_int(acc[i]));
    } else {
      atomicAdd(p.out_sum + c0 + i, static_cast<double>(acc[i]));
    }
  }
}

// general [N][C][inner]: one workgroup per (channel, slab of n); threads stride over inner
template <int ST>
__global__ __launch_bounds__(256) void channel_metric_general_kernel(const MetricParams p) {
  using S = Storage<ST>;
  using E = typename S::elem_t;
  const int64_t c = blockIdx.x;
  const int64_t n_begin = static_cast<int64_t>(blockIdx.y) * p.rows_per_block;
  const int64_t n_end = min(p.N, n_begin + p.rows_per_block);
  const E* base = static_cast<const E*>(p.x);
  float acc = 0.0f;
  for (int64_t n = n_begin; n < n_end; ++n) {
    const E* row = base + (n * p.C + c) * p.inner;
    for (int64_t i = threadIdx.x; i < p.inner; i += 256) {
      const float t = metric_term(S::load(row[i]), p.mode);
      acc = p.mode == kMetricAbsMax ? ((t > acc || t != t) ? t : acc) : acc + t;
    }
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const float a = red[threadIdx.x], b = red[threadIdx.x + off];
      red[threadIdx.x] = p.mode == kMetricAbsMax ? ((b > a || b != b) ? b : a) : a + b;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (p.mode == kMetricAbsMax) {
      atomicMax(reinterpret_cast<int*>(p.out_max) + c, __float_as_int(red[0]));
    } else {
      atomicAdd(p.out_sum + c, static_cast<double>(red[0]));
    }
  }
}

struct ScaleParams {
  const void* x;  // [outer][C][inner]
  void* y;
  const float* scale;  // [C]
  int64_t outer, C, inner, scaled_channels;
  int32_t divide;
};

template <int ST, int OUT_ST>
__global__ __launch_bounds__(256) void scale_channels_kernel(const ScaleParams p) {
  using S = Storage<ST>;
  using O = Storage<OUT_ST>;
  const int64_t total = p.outer * p.C * p.inner;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t c = (i / p.inner) % p.C;
    float v = S::load(static_cast<const typename S::elem_t*>(p.x)[i]);
    if (c < p.scaled_channels) v = p.divide ? v / p.scale[c] : v * p.scale[c];
    static_cast<typename O::elem_t*>(p.y)[i] = O::store(v);
  }
}

}  // namespace dcq

using namespace dcq;

extern "C" {

int dcq_channel_metric(const void* x, int32_t storage, int64_t N, int64_t C, int64_t inner, int32_t mode, float* out_max,
                       double* out_sum, dcq_stream_t stream) {
  DCQ_REQUIRE(N >= 0 && C > 0 && inner > 0, DCQ_ERR_ARGUMENT, "bad sizes");
  DCQ_REQUIRE(mode >= kMetricAbsMax && mode <= kMetricSquareSum, DCQ_ERR_ARGUMENT, "bad mode %d", mode);
  DCQ_REQUIRE(mode == kMetricAbsMax ? out_max != nullptr : out_sum != nullptr, DCQ_ERR_ARGUMENT, "NULL output");
  if (N == 0) return DCQ_OK;
  DCQ_REQUIRE(x != nullptr, DCQ_ERR_ARGUMENT, "NULL input");
  DCQ_REQUIRE(storage == DCQ_F32 || storage == DCQ_F16 || storage == DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage");
  MetricParams p{};
  p.x = x;
  p.N = N;
  p.C = C;
  p.inner = inner;
  p.mode = mode;
  p.out_max = out_max;
  p.out_sum = out_sum;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (inner == 1) {
    const int vec = storage == DCQ_F32 ? 4 : 8;
    const bool vec_ok = C % vec == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    const int per_thread = vec_ok ? vec : 1;
    const int64_t col_blocks = (C + 256 * per_thread - 1) / (256 * per_thread);
    int64_t row_blocks = (2048 + col_blocks - 1) / col_blocks;  // ~8 workgroups per compute unit
    if (row_blocks > (N + 63) / 64) row_blocks = (N + 63) / 64;
    if (row_blocks < 1) row_blocks = 1;
    if (row_blocks > 65535) row_blocks = 65535;
    p.rows_per_block = (N + row_blocks - 1) / row_blocks;
    const dim3 grid(static_cast<unsigned>(col_blocks), static_cast<unsigned>((N + p.rows_per_block - 1) / p.rows_per_block));
#define DCQ_METRIC(ST)                                                                                   \
  do {                                                                                                   \
    if (vec_ok) {                                                                                        \
      hipLaunchKernelGGL((channel_metric_rows_kernel<ST, Storage<ST>::kVec>), grid, dim3(256), 0, s, p); \
    } else {                                                                                             \
      hipLaunchKernelGGL((channel_metric_rows_kernel<ST, 1>), grid, dim3(256), 0, s, p);                 \
    }                                                                                                    \
  } while (0)
    if (storage == DCQ_F32) DCQ_METRIC(DCQ_F32);
    else if (storage == DCQ_F16) DCQ_METRIC(DCQ_F16);
    else DCQ_METRIC(DCQ_BF16);
#undef DCQ_METRIC
  } else {
    DCQ_REQUIRE(C <= 0x7fffffff, DCQ_ERR_UNSUPPORTED, "too many channels");
    int64_t n_blocks = (4096 + C - 1) / C;
    if (n_blocks > N) n_blocks = N;
    if (n_blocks > 65535) n_blocks = 65535;
    p.rows_per_block = (N + n_blocks - 1) / n_blocks;
    const dim3 grid(static_cast<unsigned>(C), static_cast<unsigned>((N + p.rows_per_block - 1) / p.rows_per_block));
    if (storage == DCQ_F32) hipLaunchKernelGGL(channel_metric_general_kernel<DCQ_F32>, grid, dim3(256), 0, s, p);
    else if (storage == DCQ_F16) hipLaunchKernelGGL(channel_metric_general_kernel<DCQ_F16>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(channel_metric_general_kernel<DCQ_BF16>, grid, dim3(256), 0, s, p);
  }
  DCQ_LAUNCH_CHECK("channel_metric kernel launch");
  return DCQ_OK;
}

int dcq_scale_channels(const void* x, int32_t storage, int64_t outer, int64_t C, int64_t inner, const float* scale,
                       int64_t scaled_channels, int32_t divide, void* y, int32_t out_storage, dcq_stream_t stream) {
  DCQ_REQUIRE(outer >= 0 && C > 0 && inner > 0 && scaled_channels >= 0 && scaled_channels <= C, DCQ_ERR_ARGUMENT, "bad sizes");
  if (outer == 0) return DCQ_OK;
  DCQ_REQUIRE(x && y && scale, DCQ_ERR_ARGUMENT, "NULL buffer");
  auto ok = [](int st) { return st == DCQ_F32 || st == DCQ_F16 || st == DCQ_BF16; };
  DCQ_REQUIRE(ok(storage) && ok(out_storage), DCQ_ERR_ARGUMENT, "bad storage");
  ScaleParams p{};
  p.x = x;
  p.y = y;
  p.scale = scale;
  p.outer = outer;
  p.C = C;
  p.inner = inner;
  p.scaled_channels = scaled_channels;
  p.divide = divide != 0;
  const int64_t total = outer * C * inner;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DCQ_SCALE(ST, OST) hipLaunchKernelGGL((scale_channels_kernel<ST, OST>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p)
#define DCQ_SCALE_OUT(ST)                              \
  do {                                                 \
    if (out_storage == DCQ_F32) DCQ_SCALE(ST, DCQ_F32);   \
    else if (out_storage == DCQ_F16) DCQ_SCALE(ST, DCQ_F16); \
    else DCQ_SCALE(ST, DCQ_BF16);                         \
  } while (0)
  if (storage == DCQ_F32) DCQ_SCALE_OUT(DCQ_F32);
  else if (storage == DCQ_F16) DCQ_SCALE_OUT(DCQ_F16);
  else DCQ_SCALE_OUT(DCQ_BF16);
#undef DCQ_SCALE_OUT
#undef DCQ_SCALE
  DCQ_LAUNCH_CHECK("scale_channels_kernel launch");
  return DCQ_OK;
}

}  // extern "C"
