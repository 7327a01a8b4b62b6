// Dense float32 linear algebra for GPTQ on the matrix cores (gfx950): a float32-MFMA GEMM, a blocked upper
// Cholesky factorisation and a blocked triangular inverse, composed into
//     U = upper factor with U^T U = H^-1        (deepcompressor/quantizer/kernel/gptq.py:215-233)
// and used for the lazy weight update between sweep blocks (gptq.py:269).
//
// v_mfma_f32_32x32x2_f32 is an exact float32 multiply-accumulate chain (no reduced-precision path), so results
// have plain float32 GEMM accuracy.
//
// How U is obtained.  With J the index reversal, factor J H J = R^T R (R upper).  Then H = V V^T with
// V = J R^T J upper triangular, H^-1 = V^-T V^-1, and by uniqueness of the Cholesky factor U = V^-1 = J R^-T J:
// one factorisation and one triangular inverse (2/3 n^3 flop) instead of the reference's factorise, invert,
// factorise again (4/3 n^3).  The row-major *upper* form keeps every big GEMM operand "k-major": the panel
// rows R[k][:] are at once the A operand (transposed) and the B operand of the trailing update.
#include "dcq_common.cuh"

namespace dcq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNb = 128;  // block size of the factorisation = GEMM tile

// =============================================================================================
// GEMM  C[M][N] = beta * C + alpha * A * B,  B row-major [K][N]; A either k-major (At[K][M]) or row-major [M][K].
// Tile 128 x 128 per workgroup of 4 waves (2 x 2 quadrants of 64 x 64), K step 32, register-staged double buffer.
// Optional structure: upper_only (skip tiles strictly below the diagonal of C), a_upper / b_upper (A or B is upper
// triangular with its origin on the diagonal: the k range of a tile is clipped accordingly).
// =============================================================================================
constexpr int kGT = 128, kGK = 32, kGLd = kGT + 4;  // LDS row pitch 132 floats: the two half-waves hit different banks
constexpr int kGemmThreads = 256;

struct GemmParams {
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  int32_t M, N, K;
  float alpha, beta;
  int32_t upper_only, a_upper, b_upper;
  int32_t tiles_m, tiles_n;
};

template <bool A_KMAJOR>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_f32_kernel(const GemmParams p) {
  extern __shared__ __align__(16) unsigned char gemm_smem[];
  float(*lds)[2][kGK][kGLd] = reinterpret_cast<float(*)[2][kGK][kGLd]>(gemm_smem);  // [buffer][A|B][k][m or n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
  if (p.upper_only && tn < tm) return;
  const int m0 = tm * kGT, n0 = tn * kGT;
  int k_begin = 0, k_end = p.K;
  if (p.a_upper) k_begin = max(k_begin, (m0 / kGK) * kGK);               // A[m][k] = 0 for k < m
  if (p.b_upper) k_end = min(k_end, ((n0 + kGT + kGK - 1) / kGK) * kGK);  // B[k][n] = 0 for k > n
  const int nk = k_end > k_begin ? (k_end - k_begin) / kGK : 0;

  float4 ra[4], rb[4];
  // staging maps.  k-major operand: piece = (k row, 4 consecutive m/n); row-major A: piece = (m row, 4 consecutive k)
  auto load_b = [&](int kt, float4* r) {
    const int k0 = k_begin + kt * kGK;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = it * kGemmThreads + tid;
      const int kr = id >> 5, c4 = (id & 31) * 4;
      const int n = n0 + c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* src = p.B + static_cast<int64_t>(k0 + kr) * p.ldb + n;
      if (n + 3 < p.N) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        if (n < p.N) v.x = src[0];
        if (n + 1 < p.N) v.y = src[1];
        if (n + 2 < p.N) v.z = src[2];
      }
      r[it] = v;
    }
  };
  auto load_a = [&](int kt, float4* r) {
    const int k0 = k_begin + kt * kGK;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = it * kGemmThreads + tid;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (A_KMAJOR) {
        const int kr = id >> 5, c4 = (id & 31) * 4;
        const int m = m0 + c4;
        const float* src = p.A + static_cast<int64_t>(k0 + kr) * p.lda + m;
        if (m + 3 < p.M) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (m < p.M) v.x = src[0];
          if (m + 1 < p.M) v.y = src[1];
          if (m + 2 < p.M) v.z = src[2];
        }
      } else {
        const int mr = id >> 3, k4 = (id & 7) * 4;
        if (m0 + mr < p.M) v = *reinterpret_cast<const float4*>(p.A + static_cast<int64_t>(m0 + mr) * p.lda + k0 + k4);
      }
      r[it] = v;
    }
  };
  auto store_ab = [&](int buf, const float4* a, const float4* b) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = it * kGemmThreads + tid;
      const int kr = id >> 5, c4 = (id & 31) * 4;
      *reinterpret_cast<float4*>(&lds[buf][1][kr][c4]) = b[it];
      if (A_KMAJOR) {
        *reinterpret_cast<float4*>(&lds[buf][0][kr][c4]) = a[it];
      } else {
        const int mr = id >> 3, k4 = (id & 7) * 4;
        lds[buf][0][k4 + 0][mr] = a[it].x;
        lds[buf][0][k4 + 1][mr] = a[it].y;
        lds[buf][0][k4 + 2][mr] = a[it].z;
        lds[buf][0][k4 + 3][mr] = a[it].w;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  if (nk > 0) {
    load_a(0, ra);
    load_b(0, rb);
    store_ab(0, ra, rb);
  }
  __syncthreads();
  const int fl = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      load_a(kt + 1, ra);
      load_b(kt + 1, rb);
    }
#pragma unroll
    for (int kk = 0; kk < kGK / 2; ++kk) {
      float af[2], bf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = lds[buf][0][2 * kk + fh][wr * 64 + m * 32 + fl];
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = lds[buf][1][2 * kk + fh][wc * 64 + n * 32 + fl];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    if (kt + 1 < nk) store_ab(buf ^ 1, ra, rb);
    __syncthreads();
  }
  const int col_in = lane & 31, row_off = 4 * (lane >> 5);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + row_off;
        const int col = n0 + wc * 64 + n * 32 + col_in;
        if (row < p.M && col < p.N) {
          float* c = p.C + static_cast<int64_t>(row) * p.ldc + col;
          const float v = p.alpha * acc[m][n][r];
          *c = p.beta == 0.0f ? v : p.beta * *c + v;
        }
      }
}

static int launch_gemm(const GemmParams& pin, bool a_kmajor, hipStream_t s) {
  GemmParams p = pin;
  if (p.M <= 0 || p.N <= 0) return DCQ_OK;
  p.tiles_m = (p.M + kGT - 1) / kGT;
  p.tiles_n = (p.N + kGT - 1) / kGT;
  const unsigned grid = static_cast<unsigned>(p.tiles_m) * static_cast<unsigned>(p.tiles_n);
  constexpr size_t lds = 2 * 2 * kGK * kGLd * sizeof(float);
  auto kern = a_kmajor ? gemm_f32_kernel<true> : gemm_f32_kernel<false>;
  int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds)),
                     "hipFuncSetAttribute(gemm LDS)");
  if (rc != DCQ_OK) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemmThreads), lds, s, p);
  DCQ_LAUNCH_CHECK("gemm_f32_kernel launch");
  return DCQ_OK;
}

// =============================================================================================
// Diagonal block: R11 = upper Cholesky factor of the 128 x 128 block (in place, strict lower part zeroed),
// Rinv = R11^-1 (upper, row-major [128][128]).  One workgroup; the block lives in LDS.
// info: 1-based global index of the first non-positive pivot (atomicMin so the earliest failure wins).
// =============================================================================================
constexpr int kPanelThreads = 256;

__global__ __launch_bounds__(kPanelThreads) void panel_factor_kernel(float* a, int64_t lda, float* rinv, int32_t* info,
                                                                      int32_t index_base) {
  extern __shared__ __align__(16) float sm[];
  float(*A)[kNb + 1] = reinterpret_cast<float(*)[kNb + 1]>(sm);                    // [128][129]
  float(*X)[kNb + 1] = reinterpret_cast<float(*)[kNb + 1]>(sm + kNb * (kNb + 1));  // inverse
  const int tid = threadIdx.x;
  for (int e = tid; e < kNb * kNb; e += kPanelThreads) {
    const int r = e / kNb, c = e % kNb;
    A[r][c] = c >= r ? a[r * lda + c] : 0.0f;
    X[r][c] = 0.0f;
  }
  __syncthreads();
  // right-looking upper Cholesky: row j is scaled, then the trailing upper triangle is updated
  for (int j = 0; j < kNb; ++j) {
    const float pivot = A[j][j];
    if (!(pivot > 0.0f)) {  // uniform: every thread reads the same pivot
      if (tid == 0) atomicMin(info, index_base + j + 1);
      return;
    }
    const float d = sqrtf(pivot);
    __syncthreads();  // everyone has read the pivot before row j is overwritten
    for (int c = j + tid; c < kNb; c += kPanelThreads) A[j][c] = c == j ? d : A[j][c] / d;
    __syncthreads();
    const int rem = kNb - 1 - j;  // trailing rows j+1 .. 127
    for (int e = tid; e < rem * rem; e += kPanelThreads) {
      const int r = j + 1 + e / rem, c = j + 1 + e % rem;
      if (c >= r) A[r][c] = A[r][c] - A[j][r] * A[j][c];
    }
    __syncthreads();
  }
  // inverse of the upper triangular factor: one column per thread, back substitution
  if (tid < kNb) {
    const int c = tid;
    X[c][c] = 1.0f / A[c][c];
    for (int r = c - 1; r >= 0; --r) {
      float acc = 0.0f;
      for (int k = r + 1; k <= c; ++k) acc = acc + A[r][k] * X[k][c];
      X[r][c] = -acc / A[r][r];
    }
  }
  __syncthreads();
  for (int e = tid; e < kNb * kNb; e += kPanelThreads) {
    const int r = e / kNb, c = e % kNb;
    a[r * lda + c] = A[r][c];
    rinv[e] = X[r][c];
  }
}

// dst[i][j] = S[n-1-i][n-1-j] for i, j < n and the identity on the padding up to np, where
// S = P H' P^T + diag_add * I, H' = H with zero diagonal entries replaced by 1, P the row gather `perm` (or identity)
__global__ void flip_pad_kernel(const float* src, int64_t n, const int64_t* perm, float diag_add, float* dst,
                                int64_t np) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j >= np) return;
  float v = i == j ? 1.0f : 0.0f;
  if (i < n && j < n) {
    const int64_t si = perm ? perm[n - 1 - i] : n - 1 - i, sj = perm ? perm[n - 1 - j] : n - 1 - j;
    v = src[si * n + sj];
    if (i == j) v = (v == 0.0f ? 1.0f : v) + diag_add;
  }
  dst[i * np + j] = v;
}

__global__ void finish_info_kernel(int32_t* info) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *info == 0x7f7f7f7f) *info = 0;
}

// U[i][j] = X[n-1-j][n-1-i]  (transpose + flip), X has leading dimension np; tiled through LDS
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* x, int64_t np, float* u, int64_t n) {
  __shared__ float tile[32][33];
  const int64_t bi = static_cast<int64_t>(blockIdx.y) * 32, bj = static_cast<int64_t>(blockIdx.x) * 32;  // block of U
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                                                // 32 x 8
  // U[bi + a][bj + b] = X[n-1-(bj+b)][n-1-(bi+a)]: read X rows (n-1-bj-b), columns (n-1-bi-a) coalesced along a
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int b = ty + 8 * k;
    const int64_t xr = n - 1 - (bj + b), xc = n - 1 - (bi + tx);
    tile[b][tx] = (xr >= 0 && xc >= 0) ? x[xr * np + xc] : 0.0f;  // tile[b][a]
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int64_t i = bi + a, j = bj + tx;
    if (i < n && j < n) u[i * n + j] = tile[tx][a];
  }
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

__global__ void copy_block_kernel(const float* src, float* dst, int64_t ldd) {
  for (int e = threadIdx.x; e < kNb * kNb; e += blockDim.x) dst[(e / kNb) * ldd + (e % kNb)] = src[e];
}

struct CholPlan {
  int64_t np;
  int nblk;
  size_t a_bytes, x_bytes, t_bytes, rinv_bytes;
};

static CholPlan plan_chol(int64_t n) {
  CholPlan c;
  c.np = (n + kNb - 1) / kNb * kNb;
  c.nblk = static_cast<int>(c.np / kNb);
  c.a_bytes = static_cast<size_t>(c.np) * c.np * 4;
  c.x_bytes = c.a_bytes;
  const int64_t half = (c.nblk + 1) / 2 * kNb;
  c.t_bytes = static_cast<size_t>(half) * half * 4;
  c.rinv_bytes = static_cast<size_t>(c.nblk) * kNb * kNb * 4;
  return c;
}

// X[lo:hi, lo:hi] = inverse of the upper triangular R[lo:hi, lo:hi] (block indices), recursively:
// X12 = -X11 * (R12 * X22).  X starts zeroed, so the lower parts of X11 / X22 read as zeros.
static int invert_upper(const float* R, float* X, float* T, const float* rinv, int64_t np, int lo, int hi,
                        hipStream_t s) {
  if (hi - lo == 1) {
    hipLaunchKernelGGL(copy_block_kernel, dim3(1), dim3(256), 0, s, rinv + static_cast<int64_t>(lo) * kNb * kNb,
                       X + (static_cast<int64_t>(lo) * kNb) * (np + 1), np);
    DCQ_LAUNCH_CHECK("copy_block_kernel launch");
    return DCQ_OK;
  }
  const int mid = lo + (hi - lo + 1) / 2;
  int rc = invert_upper(R, X, T, rinv, np, lo, mid, s);
  if (rc != DCQ_OK) return rc;
  rc = invert_upper(R, X, T, rinv, np, mid, hi, s);
  if (rc != DCQ_OK) return rc;
  const int m1 = (mid - lo) * kNb, m2 = (hi - mid) * kNb;
  const int64_t o1 = static_cast<int64_t>(lo) * kNb, o2 = static_cast<int64_t>(mid) * kNb;
  GemmParams g{};
  // T[m1][m2] = R12 * X22      (B = X22 upper triangular)
  g.A = R + o1 * np + o2;
  g.lda = np;
  g.B = X + o2 * np + o2;
  g.ldb = np;
  g.C = T;
  g.ldc = m2;
  g.M = m1;
  g.N = m2;
  g.K = m2;
  g.alpha = 1.0f;
  g.beta = 0.0f;
  g.b_upper = 1;
  rc = launch_gemm(g, false, s);
  if (rc != DCQ_OK) return rc;
  // X12 = -X11 * T             (A = X11 upper triangular)
  GemmParams h{};
  h.A = X + o1 * np + o1;
  h.lda = np;
  h.B = T;
  h.ldb = m2;
  h.C = X + o1 * np + o2;
  h.ldc = np;
  h.M = m1;
  h.N = m2;
  h.K = m1;
  h.alpha = -1.0f;
  h.beta = 0.0f;
  h.a_upper = 1;
  return launch_gemm(h, false, s);
}

}  // namespace dcq

using namespace dcq;

extern "C" {

int dcq_gemm_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, float alpha, float beta, dcq_stream_t stream) {
  DCQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (M == 0 || N == 0) return DCQ_OK;
  DCQ_REQUIRE(A && B && C, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(K % kGK == 0, DCQ_ERR_UNSUPPORTED, "K must be a multiple of %d (got %lld)", kGK, (long long)K);
  DCQ_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), DCQ_ERR_UNSUPPORTED, "size too large");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  DCQ_REQUIRE(al(A) && al(B) && lda % 4 == 0 && ldb % 4 == 0, DCQ_ERR_ARGUMENT,
              "A, B must be 16-byte aligned with leading dimensions that are multiples of 4");
  GemmParams g{};
  g.A = A;
  g.lda = lda;
  g.B = B;
  g.ldb = ldb;
  g.C = C;
  g.ldc = ldc;
  g.M = static_cast<int32_t>(M);
  g.N = static_cast<int32_t>(N);
  g.K = static_cast<int32_t>(K);
  g.alpha = alpha;
  g.beta = beta;
  return launch_gemm(g, a_kmajor != 0, static_cast<hipStream_t>(stream));
}

int64_t dcq_cholesky_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const CholPlan c = plan_chol(n);
  return static_cast<int64_t>(c.a_bytes + c.x_bytes + c.t_bytes + c.rinv_bytes);
}

int dcq_cholesky_inverse_upper(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U, void* workspace,
                               int64_t workspace_bytes, int32_t* info, dcq_stream_t stream) {
  DCQ_REQUIRE(n > 0 && n < (1 << 20), DCQ_ERR_ARGUMENT, "bad n");
  DCQ_REQUIRE(H && U && workspace && info, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DCQ_ERR_ARGUMENT, "workspace must be 256-byte aligned");
  DCQ_REQUIRE(workspace_bytes >= dcq_cholesky_workspace_bytes(n), DCQ_ERR_ARGUMENT, "workspace too small");
  const CholPlan c = plan_chol(n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  float* A = reinterpret_cast<float*>(ws);
  float* X = reinterpret_cast<float*>(ws + c.a_bytes);
  float* T = reinterpret_cast<float*>(ws + c.a_bytes + c.x_bytes);
  float* rinv = reinterpret_cast<float*>(ws + c.a_bytes + c.x_bytes + c.t_bytes);
  const int64_t np = c.np;
  int rc = hip_check(hipMemsetAsync(info, 0x7f, 4, s), "hipMemsetAsync(info)");  // 0x7f7f7f7f = "no failure yet"
  if (rc != DCQ_OK) return rc;
  hipLaunchKernelGGL(flip_pad_kernel, dim3(static_cast<unsigned>((np + 255) / 256), static_cast<unsigned>(np)), dim3(256),
                     0, s, H, n, perm, diag_add, A, np);
  DCQ_LAUNCH_CHECK("flip_pad_kernel launch");
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, s, X, np * np, 0.0f);
  DCQ_LAUNCH_CHECK("fill_kernel launch");
  const size_t panel_lds = 2 * kNb * (kNb + 1) * 4;
  rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_factor_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(panel_lds)),
                 "hipFuncSetAttribute(panel LDS)");
  if (rc != DCQ_OK) return rc;
  for (int kb = 0; kb < c.nblk; ++kb) {
    float* a11 = A + (static_cast<int64_t>(kb) * kNb) * (np + 1);
    float* rk = rinv + static_cast<int64_t>(kb) * kNb * kNb;
    hipLaunchKernelGGL(panel_factor_kernel, dim3(1), dim3(kPanelThreads), panel_lds, s, a11, np, rk, info, kb * kNb);
    DCQ_LAUNCH_CHECK("panel_factor_kernel launch");
    const int rest = static_cast<int>(np - (kb + 1) * kNb);
    if (rest <= 0) break;
    float* a12 = a11 + kNb;
    // R12 = R11^-T * A12, in place:  A'(m, k) = Rinv11[k][m]  (k-major operand = Rinv11 as stored)
    GemmParams g{};
    g.A = rk;
    g.lda = kNb;
    g.B = a12;
    g.ldb = np;
    g.C = a12;
    g.ldc = np;
    g.M = kNb;
    g.N = rest;
    g.K = kNb;
    g.alpha = 1.0f;
    g.beta = 0.0f;
    rc = launch_gemm(g, true, s);
    if (rc != DCQ_OK) return rc;
    // A22 -= R12^T * R12 on the upper triangle: both operands are the panel rows (k-major)
    GemmParams t{};
    t.A = a12;
    t.lda = np;
    t.B = a12;
    t.ldb = np;
    t.C = a11 + static_cast<int64_t>(kNb) * (np + 1);
    t.ldc = np;
    t.M = rest;
    t.N = rest;
    t.K = kNb;
    t.alpha = -1.0f;
    t.beta = 1.0f;
    t.upper_only = 1;
    rc = launch_gemm(t, true, s);
    if (rc != DCQ_OK) return rc;
  }
  rc = invert_upper(A, X, T, rinv, np, 0, c.nblk, s);
  if (rc != DCQ_OK) return rc;
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(static_cast<unsigned>((n + 31) / 32), static_cast<unsigned>((n + 31) / 32)),
                     dim3(256), 0, s, X, np, U, n);
  DCQ_LAUNCH_CHECK("flip_transpose_kernel launch");
  hipLaunchKernelGGL(finish_info_kernel, dim3(1), dim3(64), 0, s, info);
  DCQ_LAUNCH_CHECK("finish_info_kernel launch");
  return DCQ_OK;
}

}  // extern "C"
