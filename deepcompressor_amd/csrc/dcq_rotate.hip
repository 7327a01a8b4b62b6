// Float64 GEMM on the matrix cores (gfx950) for the rotation / Hadamard transforms of weights
// (deepcompressor/calib/rotate.py:66-133, utils/math/hadamard.py:48-61).
//
// The reference widens the weight to float64, multiplies by a float64 rotation and rounds back with
// Tensor.to(dtype).  Here the widening happens while the operand tile is staged into LDS, the products are
// accumulated by v_mfma_f64_16x16x4_f64 (an exact float64 fma chain), and the epilogue rounds double -> float ->
// storage exactly as the reference's cast does.  Operand storage types are run-time parameters (uniform
// branches): the float64 matrix instruction takes 64 cycles, staging is not the bottleneck.
#include <atomic>

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


// ---------------------------------------------------------------------------------------------
// Inverse Cholesky factor of a small float64 Gram matrix, for Cholesky QR (x <- x R^-1 with R^T R = x^T x): one
// workgroup, the matrix in LDS.  Cholesky elimination on [S | I], S = (G + G^T) / 2: after pivot j row j of the left
// part is row j of R and row j of the right part is row j of R^-T, and neither is read again -- only row j of R^-T
// is kept (it is column j of the output X = R^-1), in the place of the pivot row.  The right part is lower
// triangular and lives in the strict lower triangle of the same k x k array (its diagonal is 1 / R[j][j]).
// Thread (slot, c) keeps column c and takes the rows j + 1 + slot, + 4, ... : one barrier per pivot.
// info: 1-based index of the first non-positive pivot, or 0 (the arithmetic then runs on NaNs, harmlessly).
// ---------------------------------------------------------------------------------------------
constexpr int kGramMax = 128;      // order of the largest matrix: 128 KiB of LDS
constexpr int kGramThreads = 512;  // 4 row slots x 128 columns

__global__ __launch_bounds__(kGramThreads) void gram_inverse_factor_kernel(const double* __restrict__ G, int64_t ldg, int k,
                                                                            double* __restrict__ X, int64_t ldx,
                                                                            int32_t* __restrict__ info) {
  extern __shared__ __align__(16) unsigned char gram_smem[];
  double* M = reinterpret_cast<double*>(gram_smem);  // [k][k]
  const int c = threadIdx.x & (kGramMax - 1), slot = threadIdx.x >> 7;
  for (int e = threadIdx.x; e < k * k; e += kGramThreads) {
    const int i = e / k, j = e - i * k;
    M[e] = j >= i ? 0.5 * (G[i * ldg + j] + G[j * ldg + i]) : 0.0;
  }
  __syncthreads();
  int bad = 0;
  double finished = 0.0;
  for (int j = 0; j < k; ++j) {
    const double djj = M[j * k + j];
    if (bad == 0 && !(djj > 0.0)) bad = j + 1;
    // 1 / sqrt(djj): the hardware's estimate and two Newton steps (the first squares its 2^-26, the second ends at
    // rounding level) instead of a square root and a division in software
    double inv = __builtin_amdgcn_rsq(djj);
    inv = inv * (1.5 - 0.5 * djj * inv * inv);
    inv = inv * (1.5 - 0.5 * djj * inv * inv);
    if (c < k) {
      // this column's entry of the scaled pivot row: R^-T[j][c] (c < j), 1 / R[j][j] (c = j), R[j][c] (c > j)
      const double rjc = c == j ? inv : M[j * k + c] * inv;
      // the finished row j - 1 of R^-T replaces pivot row j - 1, which nobody reads any more (not row j of this
      // step: the other slots are still reading it; and not global memory: a barrier would wait for the stores)
      if (slot == 0 && j > 0) M[(j - 1) * k + c] = finished;
      finished = rjc;
      // Rows j + 1 + slot, + 4, ... up to `last`: a column of the right part (c <= j) takes every row below the pivot
      // (its entry in row i, column j starts as the zero the lower triangle was loaded with), a column of the left
      // part the rows down to its diagonal; what lies between, the right part's zeros of later pivots, stays.  Eight
      // rows at a time, all LDS reads of a batch ahead of its arithmetic and writes (the compiler cannot tell that
      // the writes leave the pivot row alone), then the rest one by one.
      const int last = c <= j ? k - 1 : c;
      int i = j + 1 + slot;
      for (; i + 28 <= last; i += 32) {
        double a[8], old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a[u] = M[j * k + i + 4 * u];  // R[j][i] before scaling
          old[u] = M[(i + 4 * u) * k + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) M[(i + 4 * u) * k + c] = old[u] - (a[u] * inv) * rjc;
      }
      for (; i <= last; i += 4) M[i * k + c] = M[i * k + c] - (M[j * k + i] * inv) * rjc;
    }
    __syncthreads();
  }
  if (slot == 0 && c < k) M[(k - 1) * k + c] = finished;
  __syncthreads();
  // X = (R^-T)^T, zeros below the diagonal: LDS rows read along c, the scattered 8-byte stores all in flight together
  for (int e = threadIdx.x; e < k * k; e += kGramThreads) {
    const int j = e / k, cc = e - j * k;
    X[cc * ldx + j] = cc <= j ? M[e] : 0.0;
  }
  if (threadIdx.x == 0) *info = bad;
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

extern "C" int dcq_gram_inverse_factor_f64(const double* G, int64_t ldg, int32_t k, double* X, int64_t ldx, int32_t* info,
                                           dcq_stream_t stream) {
  using namespace dcq;
  DCQ_REQUIRE(G && X && info, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(k >= 1 && ldg >= k && ldx >= k, DCQ_ERR_ARGUMENT, "bad sizes");
  DCQ_REQUIRE(k <= kGramMax, DCQ_ERR_UNSUPPORTED, "order %d: at most %d (the matrix is kept in LDS)", k, kGramMax);
  const size_t lds = static_cast<size_t>(k) * k * sizeof(double);
  static std::atomic<bool> configured[64];
  int dev = 0;
  int rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
  if (rc != DCQ_OK) return rc;
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_inverse_factor_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kGramMax * kGramMax * 8),
                   "hipFuncSetAttribute(gram LDS)");
    if (rc != DCQ_OK) return rc;
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  hipLaunchKernelGGL(gram_inverse_factor_kernel, dim3(1), dim3(kGramThreads), lds, static_cast<hipStream_t>(stream), G, ldg,
                     k, X, ldx, info);
  DCQ_LAUNCH_CHECK("gram_inverse_factor_kernel launch");
  return DCQ_OK;
}
