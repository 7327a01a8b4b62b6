// Float64 GEMM on the matrix cores (gfx950) for the rotation / Hadamard transforms of weights
// (deepcompressor/calib/rotate.py:66-133, utils/math/hadamard.py:48-61).
//
// The reference widens the weight to float64, multiplies by a float64 rotation and rounds back with
// Tensor.to(dtype).  Here the widening happens while the operand tile is staged into LDS, the products are
// accumulated by v_mfma_f64_16x16x4_f64 (an exact float64 fma chain), and the epilogue rounds double -> float ->
// storage exactly as the reference's cast does.  Operand storage types are run-time parameters (uniform
// branches): the float64 matrix instruction takes 64 cycles, staging is not the bottleneck.
#include "dcq_common.cuh"

namespace dcq {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int kDT = 64;        // tile side
constexpr int kDK = 16;        // K step
constexpr int kDLd = kDT + 16; // LDS row pitch (doubles): consecutive k rows start 32 banks apart
constexpr int kDThreads = 256;

struct Gemm64Params {
  const void* A;
  const void* B;
  void* C;
  int32_t a_storage, b_storage, c_storage;
  int32_t a_kmajor;
  int64_t lda, ldb, ldc;
  int64_t a_bstride, b_bstride, c_bstride;
  int32_t M, N, K;
  int32_t tiles_m, tiles_n;
  int32_t vec_a, vec_b;  // whole aligned 4-element pieces may be fetched with one access
};

__device__ __forceinline__ double load_as_f64(const void* base, int storage, int64_t idx) {
  switch (storage) {
    case DCQ_F64:
      return static_cast<const double*>(base)[idx];
    case DCQ_F32:
      return static_cast<double>(static_cast<const float*>(base)[idx]);
    case DCQ_F16:
      return static_cast<double>(f16_to_f32(static_cast<const uint16_t*>(base)[idx]));
    default:
      return static_cast<double>(bf16_to_f32(static_cast<const uint16_t*>(base)[idx]));
  }
}

// 4 consecutive elements base[row * ld + col .. col + 3] as doubles; zero where row >= rows or col + e >= cols
__device__ __forceinline__ void load4_f64(const void* base, int storage, int64_t ld, int row, int col, int rows, int cols,
                                          bool vec, double out[4]) {
  const bool row_ok = row < rows;
  const int64_t r = row_ok ? row : rows - 1;
  if (vec && col + 3 < cols) {
    const int64_t idx = r * ld + col;
    if (storage == DCQ_F64) {
      const double2 a = *reinterpret_cast<const double2*>(static_cast<const double*>(base) + idx);
      const double2 b = *reinterpret_cast<const double2*>(static_cast<const double*>(base) + idx + 2);
      out[0] = a.x, out[1] = a.y, out[2] = b.x, out[3] = b.y;
    } else if (storage == DCQ_F32) {
      const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
      out[0] = v.x, out[1] = v.y, out[2] = v.z, out[3] = v.w;
    } else {
      const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + idx);
      const uint16_t h[4] = {static_cast<uint16_t>(v.x & 0xffffu), static_cast<uint16_t>(v.x >> 16),
                             static_cast<uint16_t>(v.y & 0xffffu), static_cast<uint16_t>(v.y >> 16)};
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = storage == DCQ_F16 ? f16_to_f32(h[e]) : bf16_to_f32(h[e]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = col + e < cols ? col + e : cols - 1;
      const double v = load_as_f64(base, storage, r * ld + c);
      out[e] = col + e < cols ? v : 0.0;
    }
  }
  if (!row_ok) out[0] = out[1] = out[2] = out[3] = 0.0;
}

__device__ __forceinline__ void store_from_f64(void* base, int storage, int64_t idx, double v) {
  switch (storage) {
    case DCQ_F64:
      static_cast<double*>(base)[idx] = v;
      break;
    case DCQ_F32:
      static_cast<float*>(base)[idx] = static_cast<float>(v);
      break;
    case DCQ_F16:
      static_cast<uint16_t*>(base)[idx] = f32_to_f16(static_cast<float>(v));
      break;
    default:
      static_cast<uint16_t*>(base)[idx] = f32_to_bf16(static_cast<float>(v));
      break;
  }
}

__global__ __launch_bounds__(kDThreads) void gemm_f64_kernel(const Gemm64Params p) {
  __shared__ __align__(16) double lds[2][2][kDK][kDLd];  // [buffer][A | B][k][m or n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // 32 x 32 quadrant of the tile
  const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
  const int m0 = tm * kDT, n0 = tn * kDT;
  const int64_t b = blockIdx.y;
  const int asz = p.a_storage == DCQ_F64 ? 8 : p.a_storage == DCQ_F32 ? 4 : 2;
  const int bsz = p.b_storage == DCQ_F64 ? 8 : p.b_storage == DCQ_F32 ? 4 : 2;
  const int csz = p.c_storage == DCQ_F64 ? 8 : p.c_storage == DCQ_F32 ? 4 : 2;
  const void* A = static_cast<const unsigned char*>(p.A) + b * p.a_bstride * asz;
  const void* B = static_cast<const unsigned char*>(p.B) + b * p.b_bstride * bsz;
  void* C = static_cast<unsigned char*>(p.C) + b * p.c_bstride * csz;
  const int nk = (p.K + kDK - 1) / kDK;

  double ra[4], rb[4];
  // staging: one 4-element piece per thread and operand.  k-major operand: (k row = tid / 16, 4 columns);
  // row-major A: (m row = tid / 4, 4 consecutive k)
  const int kr = tid >> 4, c4 = (tid & 15) * 4;
  const int mr = tid >> 2, k4 = (tid & 3) * 4;
  auto load_ab = [&](int kt) {
    const int k0 = kt * kDK;
    load4_f64(B, p.b_storage, p.ldb, k0 + kr, n0 + c4, p.K, p.N, p.vec_b != 0, rb);
    if (p.a_kmajor) {
      load4_f64(A, p.a_storage, p.lda, k0 + kr, m0 + c4, p.K, p.M, p.vec_a != 0, ra);
    } else {
      load4_f64(A, p.a_storage, p.lda, m0 + mr, k0 + k4, p.M, p.K, p.vec_a != 0, ra);
    }
  };
  auto store_ab = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) lds[buf][1][kr][c4 + e] = rb[e];
    if (p.a_kmajor) {
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[buf][0][kr][c4 + e] = ra[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[buf][0][k4 + e][mr] = ra[e];
    }
  };

  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};

  load_ab(0);
  store_ab(0);
  __syncthreads();
  const int fl = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_ab(kt + 1);
#pragma unroll
    for (int kk = 0; kk < kDK / 4; ++kk) {
      double af[2], bf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = lds[buf][0][4 * kk + fk][wr * 32 + m * 16 + fl];
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = lds[buf][1][4 * kk + fk][wc * 32 + n * 16 + fl];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    if (kt + 1 < nk) store_ab(buf ^ 1);
    __syncthreads();
  }
  // C/D layout of the float64 instruction: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + m * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wc * 32 + n * 16 + (lane & 15);
        if (row < p.M && col < p.N) store_from_f64(C, p.c_storage, static_cast<int64_t>(row) * p.ldc + col, acc[m][n][r]);
      }
}

}  // namespace dcq

using namespace dcq;

extern "C" int dcq_gemm_f64(const void* A, int32_t a_storage, int64_t lda, int32_t a_kmajor, int64_t a_batch_stride,
                            const void* B, int32_t b_storage, int64_t ldb, int64_t b_batch_stride, void* C,
                            int32_t c_storage, int64_t ldc, int64_t c_batch_stride, int64_t M, int64_t N, int64_t K,
                            int64_t batch, dcq_stream_t stream) {
  auto storage_ok = [](int s) { return s == DCQ_F64 || s == DCQ_F32 || s == DCQ_F16 || s == DCQ_BF16; };
  DCQ_REQUIRE(storage_ok(a_storage) && storage_ok(b_storage) && storage_ok(c_storage), DCQ_ERR_ARGUMENT, "bad storage type");
  DCQ_REQUIRE(M >= 0 && N >= 0 && K >= 1 && batch >= 0, DCQ_ERR_ARGUMENT, "bad sizes");
  if (M == 0 || N == 0 || batch == 0) return DCQ_OK;
  DCQ_REQUIRE(A && B && C, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && batch < 65536, DCQ_ERR_UNSUPPORTED, "size too large");
  Gemm64Params p{};
  p.A = A;
  p.B = B;
  p.C = C;
  p.a_storage = a_storage;
  p.b_storage = b_storage;
  p.c_storage = c_storage;
  p.a_kmajor = a_kmajor != 0;
  p.lda = lda;
  p.ldb = ldb;
  p.ldc = ldc;
  p.a_bstride = a_batch_stride;
  p.b_bstride = b_batch_stride;
  p.c_bstride = c_batch_stride;
  p.M = static_cast<int32_t>(M);
  p.N = static_cast<int32_t>(N);
  p.K = static_cast<int32_t>(K);
  p.tiles_m = static_cast<int32_t>((M + kDT - 1) / kDT);
  p.tiles_n = static_cast<int32_t>((N + kDT - 1) / kDT);
  // one access per 4-element piece needs 4-element alignment of base, leading dimension and batch stride
  // (4 elements = 8 B for 16-bit types; 16-B alignment for the wider types' vector loads)
  auto vec_ok = [](const void* ptr, int64_t ld, int64_t bstride, int storage) {
    const uintptr_t need = storage == DCQ_F64 || storage == DCQ_F32 ? 15u : 7u;
    return (reinterpret_cast<uintptr_t>(ptr) & need) == 0 && ld % 4 == 0 && bstride % 4 == 0;
  };
  p.vec_a = vec_ok(A, lda, a_batch_stride, a_storage);
  p.vec_b = vec_ok(B, ldb, b_batch_stride, b_storage);
  const dim3 grid(static_cast<unsigned>(p.tiles_m) * static_cast<unsigned>(p.tiles_n), static_cast<unsigned>(batch));
  hipLaunchKernelGGL(gemm_f64_kernel, grid, dim3(kDThreads), 0, static_cast<hipStream_t>(stream), p);
  DCQ_LAUNCH_CHECK("gemm_f64_kernel launch");
  return DCQ_OK;
}
