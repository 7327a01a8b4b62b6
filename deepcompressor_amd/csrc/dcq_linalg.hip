// Dense float32 linear algebra for GPTQ on the matrix cores (gfx950): a float32-MFMA GEMM, a blocked upper
// Cholesky factorisation and a blocked triangular inverse, composed into
//     U = upper factor with U^T U = H^-1        (deepcompressor/quantizer/kernel/gptq.py:215-233)
// and used for the lazy weight update between sweep blocks (gptq.py:269).
//
// Small and medium products run on v_mfma_f32_32x32x2_f32, an exact float32 multiply-accumulate chain; large ones
// run on the 16-bit matrix pipe (gemm_split_kernel) with every float32 operand split into terms: two float16 terms
// of the value scaled by a power of two per row / column (three of the four partial products), or, where that path's
// magnitude pass cannot run, three bfloat16 terms (six of the nine partial products).  What is left out is of the
// size of a float32 rounding: all three have float32 GEMM accuracy.
//
// How U is obtained.  With J the index reversal, factor J H J = R^T R (R upper).  Then H = V V^T with
// V = J R^T J upper triangular, H^-1 = V^-T V^-1, and by uniqueness of the Cholesky factor U = V^-1 = J R^-T J:
// one factorisation and one triangular inverse (2/3 n^3 flop) instead of the reference's factorise, invert,
// factorise again (4/3 n^3).  The row-major *upper* form keeps every big GEMM operand "k-major": the panel
// rows R[k][:] are at once the A operand (transposed) and the B operand of the trailing update.
#include "dcq_common.cuh"

#include <atomic>
#include <mutex>

namespace dcq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNb = 128;  // block size of the factorisation = GEMM tile

// =============================================================================================
// GEMM  C[M][N] = beta * C + alpha * A * B,  B row-major [K][N]; A either k-major (At[K][M]) or row-major [M][K].
// Tile 128 x 128 (or 64 x 64 for small problems) per workgroup of 4 waves, K step 16 (32 for the small tile), register-staged double buffer.
// Optional structure: upper_only (skip tiles strictly below the diagonal of C) and triangular operands, for which
// the k range of a tile is clipped to the part that can be non-zero.
// =============================================================================================
constexpr int kGKMin = 16;  // K must be a multiple of every K step in use
template <int T>
struct GemmStep {
  // K step per tile size: 128-tiles run 4 workgroups per CU on a short step; 64-tiles (latency-bound small
  // problems) take fewer, longer steps
  static constexpr int value = T == 128 ? 16 : 32;
};
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
  int32_t upper_only;  // skip the tiles strictly below the diagonal of C
  // triangular operands: A(m, k) = 0 for k < m / for k > m; B[k][n] = 0 for k > n (origins on the diagonal)
  int32_t k_above_m, k_below_m, k_below_n;
  int32_t tiles_m, tiles_n;
  int32_t vec_a, vec_b, vec_c;  // 16-byte accesses allowed (pointer and leading dimension aligned)
  // two-term half-precision path only: bit patterns of max |A(m, :)| per row of C and max |B[:][n]| per column of C
  const unsigned* a_absmax;
  const unsigned* b_absmax;
  // the operands as float16 term planes, one image per 128-wide tile and K step (split_planes_kernel)
  const unsigned char* a_planes;
  const unsigned char* b_planes;
};

// One 16-byte piece of a row: `FAST` pieces are whole and aligned; otherwise every element is guarded by `limit`
// (clamped address + select, so no lane-divergent branch and no wait between the loads).
template <bool FAST>
__device__ __forceinline__ float4 load_piece(const float* row, int col, int limit) {
  if (FAST) return *reinterpret_cast<const float4*>(row + col);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t = row[min(col + e, limit - 1)];
    v[e] = col + e < limit ? t : 0.0f;
  }
  return make_float4(v[0], v[1], v[2], v[3]);
}

// One T x T tile of C by 4 waves (2 x 2 quadrants of T/2 x T/2, each T/64 x T/64 matrix-instruction tiles).
// T = 128 for throughput; T = 64 spreads small problems over 4x as many compute units (latency-bound regime).
// `tid` (0..255) is the thread's index in the workgroup.
template <bool A_KMAJOR, bool FAST, int T>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, float* lds, int m0, int n0, int k_begin, int nk, int tid) {
  constexpr int kGK = GemmStep<T>::value;
  constexpr int LD = T + 4;              // LDS row pitch: the two half-waves of a fragment read hit different banks
  constexpr int IMG = kGK * LD;          // one operand image [k][m or n]
  constexpr int PIECES = kGK * T / 4 / kGemmThreads;  // 16-byte pieces per thread, operand and k-step
  constexpr int KP = kGK / 4;            // pieces per row of a row-major A tile
  constexpr int F = T / 64;              // 32 x 32 tiles per wave and dimension
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float4 ra[PIECES], rb[PIECES];
  // staging maps.  k-major operand: piece = (k row, 4 consecutive m/n); row-major A: piece = (m row, 4 consecutive k)
  auto load_ab = [&](int kt) {
    const int k0 = k_begin + kt * kGK;
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int id = it * kGemmThreads + tid;
      const int kr = id / (T / 4), c4 = (id % (T / 4)) * 4;
      rb[it] = load_piece<FAST>(p.B + static_cast<int64_t>(k0 + kr) * p.ldb, n0 + c4, p.N);
      if (A_KMAJOR) {
        ra[it] = load_piece<FAST>(p.A + static_cast<int64_t>(k0 + kr) * p.lda, m0 + c4, p.M);
      } else {
        const int mr = id / KP, k4 = (id % KP) * 4;
        const int m = m0 + mr;
        const float* row = p.A + static_cast<int64_t>(min(m, p.M - 1)) * p.lda + k0;
        // K is a multiple of the step: only alignment can be missing.  Rows past M are read from row M - 1 and
        // zeroed when they are written to LDS, after the arithmetic of the current step: a select on the loaded
        // value here would make the wave wait for the load before its matrix instructions.
        ra[it] = load_piece<FAST>(row, k4, kGK);
      }
    }
  };
  auto store_ab = [&](int buf) {
    float* la = lds + buf * 2 * IMG;
    float* lb = la + IMG;
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int id = it * kGemmThreads + tid;
      const int kr = id / (T / 4), c4 = (id % (T / 4)) * 4;
      *reinterpret_cast<float4*>(lb + kr * LD + c4) = rb[it];
      if (A_KMAJOR) {
        *reinterpret_cast<float4*>(la + kr * LD + c4) = ra[it];
      } else {
        const int mr = id / KP, k4 = (id % KP) * 4;
        const bool live = m0 + mr < p.M;
        la[(k4 + 0) * LD + mr] = live ? ra[it].x : 0.0f;
        la[(k4 + 1) * LD + mr] = live ? ra[it].y : 0.0f;
        la[(k4 + 2) * LD + mr] = live ? ra[it].z : 0.0f;
        la[(k4 + 3) * LD + mr] = live ? ra[it].w : 0.0f;
      }
    }
  };

  f32x16 acc[F][F];
#pragma unroll
  for (int m = 0; m < F; ++m)
#pragma unroll
    for (int n = 0; n < F; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  if (nk > 0) {
    load_ab(0);
    store_ab(0);
  }
  __syncthreads();
  const int fl = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const float* la = lds + (kt & 1) * 2 * IMG + fh * LD + wr * (T / 2) + fl;
    const float* lb = lds + (kt & 1) * 2 * IMG + IMG + fh * LD + wc * (T / 2) + fl;
    if (kt + 1 < nk) load_ab(kt + 1);
#pragma unroll
    for (int kk = 0; kk < kGK / 2; ++kk) {
      float af[F], bf[F];
#pragma unroll
      for (int m = 0; m < F; ++m) af[m] = la[2 * kk * LD + m * 32];
#pragma unroll
      for (int n = 0; n < F; ++n) bf[n] = lb[2 * kk * LD + n * 32];
#pragma unroll
      for (int m = 0; m < F; ++m)
#pragma unroll
        for (int n = 0; n < F; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    if (kt + 1 < nk) store_ab((kt + 1) & 1);
    __syncthreads();
  }
  const int col_in = lane & 31, row_off = 4 * (lane >> 5);
  const bool inside = m0 + T <= p.M && n0 + T <= p.N;  // uniform
  // Whole tiles take a branch-free path: the 16 values of C that a 32 x 32 block accumulates into are all requested
  // before the first is used, and the 16 results stand in registers of their own before the first store.  Taken one by
  // one behind a bounds check each, every read waits for the store before it (loads and stores share one counter) --
  // sixteen round trips to memory at the end of a product that may be eight K steps long.
#pragma unroll
  for (int m = 0; m < F; ++m)
#pragma unroll
    for (int n = 0; n < F; ++n) {
      const int col = n0 + wc * (T / 2) + n * 32 + col_in;
      const int row0 = m0 + wr * (T / 2) + m * 32 + row_off;
      float* c0 = p.C + static_cast<int64_t>(row0) * p.ldc + col;
      if (inside) {
        float out[16];
        if (p.beta != 0.0f) {  // uniform
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = c0[static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * p.ldc];
#pragma unroll
          for (int r = 0; r < 16; ++r) out[r] = p.beta * old[r] + p.alpha * acc[m][n][r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) out[r] = p.alpha * acc[m][n][r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * p.ldc] = out[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int up = (r & 3) + 8 * (r >> 2);
          if (row0 + up < p.M && col < p.N) {
            float* c = c0 + static_cast<int64_t>(up) * p.ldc;
            const float v = p.alpha * acc[m][n][r];
            *c = p.beta == 0.0f ? v : p.beta * *c + v;
          }
        }
      }
    }
}

template <bool A_KMAJOR, int T>
__global__ __launch_bounds__(kGemmThreads, 4) void gemm_f32_kernel(const GemmParams p) {
  extern __shared__ __align__(16) unsigned char gemm_smem[];
  float* lds = reinterpret_cast<float*>(gemm_smem);  // [buffer][A | B][k][m or n]
  int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
  // triangular operands: the tiles with the longest reductions are dispatched first (largest m when k <= m, largest
  // n when k <= n; k >= m already has them first), so the launch does not end on a few full-length tiles
  if (p.k_below_m) tm = p.tiles_m - 1 - tm;
  if (p.k_below_n) tn = p.tiles_n - 1 - tn;
  if (p.upper_only && tn < tm) return;
  constexpr int kGK = GemmStep<T>::value;
  const int m0 = tm * T, n0 = tn * T;
  int k_begin = 0, k_end = p.K;
  if (p.k_above_m) k_begin = max(k_begin, m0 / kGK * kGK);
  if (p.k_below_m) k_end = min(k_end, m0 + T);
  if (p.k_below_n) k_end = min(k_end, n0 + T);
  const int nk = k_end > k_begin ? (k_end - k_begin) / kGK : 0;
  // whole, aligned 16-byte pieces everywhere in this tile?  (uniform over the workgroup)
  const bool fast = p.vec_a && p.vec_b && n0 + T <= p.N && (!A_KMAJOR || m0 + T <= p.M);
  if (fast) {
    gemm_tile<A_KMAJOR, true, T>(p, lds, m0, n0, k_begin, nk, threadIdx.x);
  } else {
    gemm_tile<A_KMAJOR, false, T>(p, lds, m0, n0, k_begin, nk, threadIdx.x);
  }
}

template <bool A_KMAJOR, int T>
static int launch_gemm_variant(const GemmParams& p, hipStream_t s) {
  constexpr size_t lds = 2 * 2 * GemmStep<T>::value * (T + 4) * sizeof(float);
  auto kern = gemm_f32_kernel<A_KMAJOR, T>;
  static std::atomic<bool> configured[64];  // (set twice by racing first calls at worst: the attribute call is idempotent)
  int dev = 0;
  int rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
  if (rc != DCQ_OK) return rc;
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)),
                   "hipFuncSetAttribute(gemm LDS)");
    if (rc != DCQ_OK) return rc;
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  const unsigned grid = static_cast<unsigned>(p.tiles_m) * static_cast<unsigned>(p.tiles_n);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemmThreads), lds, s, p);
  DCQ_LAUNCH_CHECK("gemm_f32_kernel launch");
  return DCQ_OK;
}

// =============================================================================================
// The same GEMM on the bf16 matrix pipe, for large problems.  A float32 value splits exactly into three bfloat16
// terms x = h + m + l (3 x 8 significant bits, each the round-to-nearest of what the previous ones left), and a
// product is accumulated in float32 as  h h' + h m' + m h' + m m' + h l' + l h' : the three terms left out are below
// 2^-23 |x x'|, the size of one float32 rounding, so the result has float32-GEMM accuracy while six bf16 matrix
// instructions (16 x the rate of a float32 one each) stand where sixteen float32 ones did.
//
// With two float16 terms (11 significant bits each plus the sign of the second): x = h + l up to 2^-22 |x|, the
// product is h h' + h l' + l h' on three matrix instructions, and the operands are first brought into float16's range,
// exactly, by a power of two per row of A and per column of B that the epilogue takes out again (half_scale_of).
//
// Tile 128 x 128 per workgroup of 4 waves, K step 32.  Waves 0-1 stage the A operand, waves 2-3 the B operand:
// float32 from global memory into registers (issued one K step ahead), split, written to LDS as three (two) planes of
// [k chunk of 8][row][8 bf16], the layout both operands of v_mfma_f32_16x16x32_bf16 read with one 16-byte access
// per lane.  LDS row s of a tile holds the tile's row (or column) 64 (s / 64) + 4 (s % 16) + (s / 16) % 4 (stored at
// row s ^ chunk, which spreads the 8-byte writes of a row-major A operand over the banks): a thread
// that loaded 4 consecutive columns of a k-major operand then writes, together with its 15 neighbours, 256
// contiguous bytes per column (no bank conflict), and in the result a lane holds 4 consecutive columns of C.
// =============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kSplitK = 32;
constexpr int kSplitChunkBytes = 128 * 16;                   // [row 128][8 bf16]
constexpr int kSplitPlaneBytes = 4 * kSplitChunkBytes;       // 4 chunks of 8 = one K step
constexpr int split_operand_bytes(int planes) { return planes * kSplitPlaneBytes; }
constexpr int split_lds_bytes(int planes) { return 2 * split_operand_bytes(planes); }  // 48 KiB (3 planes) / 32 KiB (2)

__device__ __forceinline__ void split_pair(f32x2 v, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
  const f32x2 r1 = v - hf;  // exact
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
  const f32x2 r2 = r1 - mf;  // exact
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// two float16 terms: h = round-to-nearest of v, l = round-to-nearest of what h left (v - h is exact in float32)
__device__ __forceinline__ void split_pair(f32x2 v, unsigned& h, unsigned& l) {
  const f16x2 hh = __builtin_convertvector(v, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const f32x2 r = v - __builtin_convertvector(hh, f32x2);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// 8 consecutive-k values of one row -> the row's 16 bytes in each of the planes
template <int PLANES>
__device__ __forceinline__ void split_store(const float (&x)[8], unsigned char* at) {
  if constexpr (PLANES == 3) {
    uint4 h, m, l;
    split_pair(f32x2{x[0], x[1]}, h.x, m.x, l.x);
    split_pair(f32x2{x[2], x[3]}, h.y, m.y, l.y);
    split_pair(f32x2{x[4], x[5]}, h.z, m.z, l.z);
    split_pair(f32x2{x[6], x[7]}, h.w, m.w, l.w);
    *reinterpret_cast<uint4*>(at) = h;
    *reinterpret_cast<uint4*>(at + kSplitPlaneBytes) = m;
    *reinterpret_cast<uint4*>(at + 2 * kSplitPlaneBytes) = l;
  } else {
    uint4 h, l;
    split_pair(f32x2{x[0], x[1]}, h.x, l.x);
    split_pair(f32x2{x[2], x[3]}, h.y, l.y);
    split_pair(f32x2{x[4], x[5]}, h.z, l.z);
    split_pair(f32x2{x[6], x[7]}, h.w, l.w);
    *reinterpret_cast<uint4*>(at) = h;
    *reinterpret_cast<uint4*>(at + kSplitPlaneBytes) = l;
  }
}

// The power of two that brings a row (or column) of largest magnitude `absmax_bits` (a float32 bit pattern) into
// [2^14, 2^15), and its reciprocal.  Zero, denormal and non-finite maxima leave the values as they are.
__device__ __forceinline__ void half_scale_of(unsigned absmax_bits, float& scale, float& inverse) {
  const int ex = static_cast<int>((absmax_bits >> 23) & 0xffu);
  int se = (ex == 0 || ex == 255) ? 127 : 268 - ex;
  se = min(max(se, 1), 253);
  scale = __builtin_bit_cast(float, static_cast<unsigned>(se) << 23);
  inverse = __builtin_bit_cast(float, static_cast<unsigned>(254 - se) << 23);
}

__device__ __forceinline__ int split_row_of(int s) { return 64 * (s >> 6) + 4 * (s & 15) + ((s >> 4) & 3); }

// Result of one 128 x 128 tile: lane (g, j) of block (i, n) holds C[64 wr + 16 g + 4 r + i][64 wc + 4 j + n], r = 0..3
// `c_old`: the tile's own values of C, [i][r] in the order below, when the caller has fetched them already
// (HAVE_OLD: whole tiles with beta != 0 only).
template <int PLANES, bool HAVE_OLD>
__device__ __forceinline__ void split_epilogue(const GemmParams& p, const f32x4 (&acc)[4][4], int m0, int n0, int wr, int wc,
                                               int g, int j, const float4 (&c_old)[4][4]) {
  const bool whole = m0 + 128 <= p.M && n0 + 128 <= p.N && p.vec_c;  // uniform
  float back_b[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // two-term path: the scales are taken out again, exactly
  if constexpr (PLANES == 2) {
    float unused;
#pragma unroll
    for (int e = 0; e < 4; ++e) half_scale_of(p.b_absmax[min(n0 + 64 * wc + 4 * j + e, p.N - 1)], unused, back_b[e]);
  }
  // (whole tiles and tiles on an edge in loops of their own, and the rows' magnitudes fetched ahead of the loop: with
  // bounds checks and loads between the stores every store waits for the one before it -- loads and stores share one
  // counter)
  if (whole) {
    unsigned a_bits[4][4];
    if constexpr (PLANES == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) a_bits[i][r] = p.a_absmax[m0 + 64 * wr + 16 * g + 4 * r + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 64 * wr + 16 * g + 4 * r + i, col = n0 + 64 * wc + 4 * j;
        float v[4];
        if constexpr (PLANES == 2) {
          float unused, back_a;
          half_scale_of(a_bits[i][r], unused, back_a);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p.alpha * ((acc[i][e][r] * back_a) * back_b[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p.alpha * acc[i][e][r];
        }
        float* c = p.C + static_cast<int64_t>(row) * p.ldc + col;
        float4 out = make_float4(v[0], v[1], v[2], v[3]);
        if (p.beta != 0.0f) {
          const float4 old = HAVE_OLD ? c_old[i][r] : *reinterpret_cast<const float4*>(c);
          out = make_float4(p.beta * old.x + v[0], p.beta * old.y + v[1], p.beta * old.z + v[2], p.beta * old.w + v[3]);
        }
        *reinterpret_cast<float4*>(c) = out;
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 64 * wr + 16 * g + 4 * r + i, col = n0 + 64 * wc + 4 * j;
      float v[4];
      if constexpr (PLANES == 2) {
        float unused, back_a;
        half_scale_of(p.a_absmax[min(row, p.M - 1)], unused, back_a);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p.alpha * ((acc[i][e][r] * back_a) * back_b[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p.alpha * acc[i][e][r];
      }
      float* c = p.C + static_cast<int64_t>(row) * p.ldc + col;
      if (row < p.M) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < p.N) c[e] = p.beta == 0.0f ? v[e] : p.beta * c[e] + v[e];
      }
    }
}

template <bool A_KMAJOR, bool FAST>
__device__ __forceinline__ void gemm_split_tile(const GemmParams& p, unsigned char* lds, int m0, int n0, int k_begin,
                                                int nk) {
  constexpr int kOperandBytes = split_operand_bytes(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const bool stages_a = tid < 128;  // uniform per wave
  const int u = tid & 127;
  // k-major operand: thread = (4 consecutive columns 4 q.., k chunk kc); row-major A: thread = piece u % 8 of 8 rows
  float4 regs[2][8];  // two sets of staging registers, used in turn: K step kt + 2 is requested while kt is multiplied
  const int q = u & 31, kc = u >> 5;
  auto load_stage = [&](int kt, const int set) __attribute__((always_inline)) {
    const int k0 = k_begin + kt * kSplitK;
    if (stages_a && !A_KMAJOR) {
      // 8 lanes read the 128 bytes of one row; a thread holds piece u % 8 of the rows 16 it + u / 8
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = min(m0 + split_row_of(16 * it + (u >> 3)), p.M - 1);
        regs[set][it] = load_piece<FAST>(p.A + static_cast<int64_t>(row) * p.lda + k0, 4 * (u & 7), kSplitK);
      }
    } else {
      const float* src = stages_a ? p.A : p.B;
      const int64_t ld = stages_a ? p.lda : p.ldb;
      const int c0 = (stages_a ? m0 : n0) + 4 * q, limit = stages_a ? p.M : p.N;
#pragma unroll
      for (int j = 0; j < 8; ++j) regs[set][j] = load_piece<FAST>(src + static_cast<int64_t>(k0 + 8 * kc + j) * ld, c0, limit);
    }
  };
  auto store_stage = [&](const int set) __attribute__((always_inline)) {
    unsigned char* img = lds + (stages_a ? 0 : kOperandBytes);
    if (stages_a && !A_KMAJOR) {
      const int ck = (u & 7) >> 1;  // the piece's k chunk; its half of the chunk: u % 2
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int srow = 16 * it + (u >> 3);
        const bool live = m0 + split_row_of(srow) < p.M;
        const float4 v = live ? regs[set][it] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        unsigned char* at = img + ck * kSplitChunkBytes + (srow ^ ck) * 16 + (u & 1) * 8;
        uint2 h, m, l;
        split_pair(f32x2{v.x, v.y}, h.x, m.x, l.x);
        split_pair(f32x2{v.z, v.w}, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(at) = h;
        *reinterpret_cast<uint2*>(at + kSplitPlaneBytes) = m;
        *reinterpret_cast<uint2*>(at + 2 * kSplitPlaneBytes) = l;
      }
    } else {
      // rows (q % 16) + 16 c + 64 (q / 16), c = 0..3; kc only flips the two lowest bits of the row
      unsigned char* at = img + kc * kSplitChunkBytes + (((q & 15) ^ kc) + 64 * (q >> 4)) * 16;
      {
        const float x[8] = {regs[set][0].x, regs[set][1].x, regs[set][2].x, regs[set][3].x,
                            regs[set][4].x, regs[set][5].x, regs[set][6].x, regs[set][7].x};
        split_store<3>(x, at);
      }
      {
        const float x[8] = {regs[set][0].y, regs[set][1].y, regs[set][2].y, regs[set][3].y,
                            regs[set][4].y, regs[set][5].y, regs[set][6].y, regs[set][7].y};
        split_store<3>(x, at + 16 * 16);
      }
      {
        const float x[8] = {regs[set][0].z, regs[set][1].z, regs[set][2].z, regs[set][3].z,
                            regs[set][4].z, regs[set][5].z, regs[set][6].z, regs[set][7].z};
        split_store<3>(x, at + 32 * 16);
      }
      {
        const float x[8] = {regs[set][0].w, regs[set][1].w, regs[set][2].w, regs[set][3].w,
                            regs[set][4].w, regs[set][5].w, regs[set][6].w, regs[set][7].w};
        split_store<3>(x, at + 48 * 16);
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  if (nk > 0) {
    load_stage(0, 0);
    if (nk > 1) load_stage(1, 1);
    store_stage(0);
  }
  __syncthreads();
  const int g = lane >> 4, j = lane & 15;
  const unsigned char* fa = lds + g * kSplitChunkBytes + (64 * wr + (j ^ g)) * 16;
  const unsigned char* fb = lds + kOperandBytes + g * kSplitChunkBytes + (64 * wc + (j ^ g)) * 16;
  auto multiply_step = [&]() __attribute__((always_inline)) {
    bf16x8 a[3][4], b[3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fa + pl * kSplitPlaneBytes + i * 256));
        b[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fb + pl * kSplitPlaneBytes + i * 256));
      }
    // smallest terms first
    constexpr int pa[6] = {0, 2, 1, 0, 1, 0}, pb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa[t]][i], b[pb[t]][n], acc[i][n], 0, 0, 0);
  };
  // one K step: multiply the image of step kt, then replace it by step kt + 1 (register set `have`); set `next` takes kt + 2
  auto step = [&](int kt, const int have, const int next) __attribute__((always_inline)) {
    if (kt + 2 < nk) load_stage(kt + 2, next);
    multiply_step();
    __syncthreads();  // every wave has read this K step's image
    if (kt + 1 < nk) store_stage(have);
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, 1, 0);
    if (kt + 1 < nk) step(kt + 1, 0, 1);
  }
  const float4 none[4][4] = {};
  split_epilogue<3, false>(p, acc, m0, n0, wr, wc, g, j, none);
}

template <bool A_KMAJOR>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_split_kernel(const GemmParams p) {
  extern __shared__ __align__(16) unsigned char gemm_smem[];
  int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
  if (p.k_below_m) tm = p.tiles_m - 1 - tm;  // longest reductions first (see gemm_f32_kernel)
  if (p.k_below_n) tn = p.tiles_n - 1 - tn;
  if (p.upper_only && tn < tm) return;
  const int m0 = tm * 128, n0 = tn * 128;
  int k_begin = 0, k_end = p.K;
  if (p.k_above_m) k_begin = max(k_begin, m0 / kSplitK * kSplitK);
  if (p.k_below_m) k_end = min(k_end, m0 + 128);
  if (p.k_below_n) k_end = min(k_end, n0 + 128);
  const int nk = k_end > k_begin ? (k_end - k_begin) / kSplitK : 0;
  const bool fast = p.vec_a && p.vec_b && n0 + 128 <= p.N && (!A_KMAJOR || m0 + 128 <= p.M);
  if (fast) {
    gemm_split_tile<A_KMAJOR, true>(p, gemm_smem, m0, n0, k_begin, nk);
  } else {
    gemm_split_tile<A_KMAJOR, false>(p, gemm_smem, m0, n0, k_begin, nk);
  }
}

template <bool A_KMAJOR>
static int launch_gemm_split(const GemmParams& p, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(p.tiles_m) * static_cast<unsigned>(p.tiles_n);
  hipLaunchKernelGGL(gemm_split_kernel<A_KMAJOR>, dim3(grid), dim3(kGemmThreads), split_lds_bytes(3), s, p);
  DCQ_LAUNCH_CHECK("gemm_split_kernel launch");
  return DCQ_OK;
}

// ---------------------------------------------------------------------------------------------
// The two-term path: the operands are split ONCE, ahead of the product.  In the three-term kernel above every
// workgroup converts the operand tiles it stages, so a value of a 7168-wide product is split 56 times, and the
// conversions share the vector pipe with the address arithmetic of the loads.  split_planes_kernel writes, for every
// 128-wide tile of an operand and every K step of 32, a 16 KiB image in the layout above (2 planes x 4 chunks of 8 x
// [row position][8 float16], rows permuted and swizzled): what a wave's lane reads as its fragment is 16 contiguous
// bytes of it.  gemm_planes_kernel copies the A images into LDS (16-byte pieces, linear), reads the B fragments from
// the images directly, and multiplies.
// ---------------------------------------------------------------------------------------------
constexpr int kImageBytes = 2 * kSplitPlaneBytes;  // one operand tile, one K step

// `clip`: 1 = the operand is zero for k beyond the end of the tile's own 128-block (and those blocks may hold anything),
// 2 = for k before its start.  Workgroup = two K steps of one tile.
template <bool KMAJOR>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t ld, int extent, int K,
                                                           const unsigned* __restrict__ absmax, int clip,
                                                           unsigned char* __restrict__ planes) {
  const int ksteps = K / kSplitK;
  const int t = blockIdx.x, ks = blockIdx.y * 2 + (threadIdx.x >> 7);
  if (ks >= ksteps) return;
  const int k0 = ks * kSplitK, c0 = t * 128;
  if (clip == 1 && k0 >= c0 + 128) return;
  if (clip == 2 && k0 + kSplitK <= c0) return;
  unsigned char* img = planes + (static_cast<int64_t>(t) * ksteps + ks) * kImageBytes;
  const int u = threadIdx.x & 127;
  float unused;
  if (KMAJOR) {
    // thread = (4 consecutive columns 4 q .., k chunk kc)
    const int q = u & 31, kc = u >> 5;
    const int col = c0 + 4 * q;
    const bool live = col < extent;  // extent is a multiple of 4
    float sc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) half_scale_of(absmax[min(col + e, extent - 1)], sc[e], unused);
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = live ? *reinterpret_cast<const float4*>(src + static_cast<int64_t>(k0 + 8 * kc + j) * ld + col)
                  : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned char* at = img + kc * kSplitChunkBytes + (((q & 15) ^ kc) + 64 * (q >> 4)) * 16;
    {
      const float x[8] = {v[0].x * sc[0], v[1].x * sc[0], v[2].x * sc[0], v[3].x * sc[0],
                          v[4].x * sc[0], v[5].x * sc[0], v[6].x * sc[0], v[7].x * sc[0]};
      split_store<2>(x, at);
    }
    {
      const float x[8] = {v[0].y * sc[1], v[1].y * sc[1], v[2].y * sc[1], v[3].y * sc[1],
                          v[4].y * sc[1], v[5].y * sc[1], v[6].y * sc[1], v[7].y * sc[1]};
      split_store<2>(x, at + 16 * 16);
    }
    {
      const float x[8] = {v[0].z * sc[2], v[1].z * sc[2], v[2].z * sc[2], v[3].z * sc[2],
                          v[4].z * sc[2], v[5].z * sc[2], v[6].z * sc[2], v[7].z * sc[2]};
      split_store<2>(x, at + 32 * 16);
    }
    {
      const float x[8] = {v[0].w * sc[3], v[1].w * sc[3], v[2].w * sc[3], v[3].w * sc[3],
                          v[4].w * sc[3], v[5].w * sc[3], v[6].w * sc[3], v[7].w * sc[3]};
      split_store<2>(x, at + 48 * 16);
    }
  } else {
    // 8 lanes read the 128 bytes of one row; a thread holds piece u % 8 of the rows at positions 16 it + u / 8
    const int ck = (u & 7) >> 1;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int srow = 16 * it + (u >> 3);
      const int row = c0 + split_row_of(srow);
      const bool live = row < extent;
      float scale;
      half_scale_of(absmax[min(row, extent - 1)], scale, unused);
      const float4 v = live ? *reinterpret_cast<const float4*>(src + static_cast<int64_t>(row) * ld + k0 + 4 * (u & 7))
                            : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      uint2 h, l;
      split_pair(f32x2{v.x * scale, v.y * scale}, h.x, l.x);
      split_pair(f32x2{v.z * scale, v.w * scale}, h.y, l.y);
      unsigned char* at = img + ck * kSplitChunkBytes + (srow ^ ck) * 16 + (u & 1) * 8;
      *reinterpret_cast<uint2*>(at) = h;
      *reinterpret_cast<uint2*>(at + kSplitPlaneBytes) = l;
    }
  }
}

constexpr int kPlanesLdsBytes = 2 * kImageBytes;  // two buffers of one A image: 32 KiB

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void planes_multiply(f32x4 (&acc)[4][4], const u32x4 (&a)[2][4], const u32x4 (&b)[2][4]) {
  constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};  // l h', h l', h h': smallest terms first
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[pa[t]][i]),
                                                           __builtin_bit_cast(f16x8, b[pb[t]][n]), acc[i][n], 0, 0, 0);
}

// past the end the loads repeat the last image; nobody uses what they deliver
__device__ __forceinline__ void planes_request_a(int image, int nk, const unsigned char* ga, u32x4 (&ra)[4]) {
  const int64_t at_image = static_cast<int64_t>(min(image, nk - 1)) * kImageBytes;
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ga + at_image + i * 4096);
}

// a wave's B fragments of one K step, straight from the image in global memory (it is in fragment order)
__device__ __forceinline__ void planes_request_b(int image, int nk, const unsigned char* gb, u32x4 (&b)[2][4]) {
  const int64_t at_image = static_cast<int64_t>(min(image, nk - 1)) * kImageBytes;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int i = 0; i < 4; ++i) b[pl][i] = *reinterpret_cast<const u32x4*>(gb + at_image + pl * kSplitPlaneBytes + i * 256);
}

// One K step of gemm_planes_kernel.  The register set `ra` holds A image kt + 1, requested two steps ago: it goes into
// the LDS buffer that image kt - 1 has left (every wave read it before the barrier that ended step kt - 1) and the set
// is requested again for image kt + 3; the A fragments of image kt come from LDS, the B fragments `b` of step kt were
// requested two steps ago and their registers are requested again for step kt + 2.
__device__ __forceinline__ void planes_step(int kt, int nk, const unsigned char* ga, const unsigned char* gb,
                                            unsigned char* lds, int tid, int frag_a, f32x4 (&acc)[4][4], u32x4 (&ra)[4],
                                            u32x4 (&b)[2][4]) {
  unsigned char* at = lds + ((kt + 1) & 1) * kImageBytes + tid * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(at + i * 4096) = ra[i];
  planes_request_a(kt + 3, nk, ga, ra);
  u32x4 a[2][4];
  const unsigned char* img = lds + (kt & 1) * kImageBytes + frag_a;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[pl][i] = *reinterpret_cast<const u32x4*>(img + pl * kSplitPlaneBytes + i * 256);
  planes_multiply(acc, a, b);
  planes_request_b(kt + 2, nk, gb, b);
  __syncthreads();
}

// The A images go through LDS (two buffers, one barrier per K step); a wave reads its B fragments from global memory,
// where the images are in fragment order already, which halves the LDS traffic and leaves room for three workgroups'
// worth of LDS.  What bounds the kernel is the matrix pipe: 192 instructions of 16 x 16 x 32 per tile and K step at
// the 32 cycles each that this chip sustains on 16-bit operands (its best dense 16-bit GEMM rate, 1.25 PFLOP/s, is
// half the nominal one) are 1536 cycles per compute unit; measured 1750, and the same within 5 % with both operands
// through LDS, with neither (all fragments from global memory, no barrier), with the fragments read one step ahead,
// with one or three images in flight, and with two tiles per 8-wave workgroup taking turns at the matrix pipe.
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_planes_kernel(const GemmParams p) {
  extern __shared__ __align__(16) unsigned char gemm_smem[];
  int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
  if (p.k_below_m) tm = p.tiles_m - 1 - tm;  // longest reductions first (see gemm_f32_kernel)
  if (p.k_below_n) tn = p.tiles_n - 1 - tn;
  if (p.upper_only && tn < tm) return;
  const int m0 = tm * 128, n0 = tn * 128;
  int k_begin = 0, k_end = p.K;
  if (p.k_above_m) k_begin = max(k_begin, m0 / kSplitK * kSplitK);
  if (p.k_below_m) k_end = min(k_end, m0 + 128);
  if (p.k_below_n) k_end = min(k_end, n0 + 128);
  const int nk = k_end > k_begin ? (k_end - k_begin) / kSplitK : 0;
  const int ksteps = p.K / kSplitK, ks0 = k_begin / kSplitK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // C += ...: the tile's values of C are requested ahead of the operands, so that the epilogue does not start with a
  // round trip to memory (a short reduction -- the sweeps' K = 512 -- is over in 16 steps)
  const bool fetch_c = p.beta != 0.0f && m0 + 128 <= p.M && n0 + 128 <= p.N && p.vec_c;  // uniform
  float4 c_old[4][4];
  if (fetch_c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        c_old[i][r] = *reinterpret_cast<const float4*>(p.C + static_cast<int64_t>(m0 + 64 * wr + 16 * g + 4 * r + i) * p.ldc +
                                                       n0 + 64 * wc + 4 * j);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nk > 0) {  // uniform over the workgroup
    const unsigned char* ga = p.a_planes + (static_cast<int64_t>(tm) * ksteps + ks0) * kImageBytes + tid * 16;
    const unsigned char* gb = p.b_planes + (static_cast<int64_t>(tn) * ksteps + ks0) * kImageBytes +
                              g * kSplitChunkBytes + (64 * wc + (j ^ g)) * 16;
    const int frag_a = g * kSplitChunkBytes + (64 * wr + (j ^ g)) * 16;
    u32x4 ra0[4], ra1[4];        // A image 2 i + s on its way to LDS in set s
    u32x4 b0[2][4], b1[2][4];    // B fragments of step 2 i + s in set s
    // the requests in the order the loop leaves them in at its end (A image kt + 1, B step kt, A image kt + 2, B step
    // kt + 1), pinned: where the order on entry differs from the one around the loop the compiler waits for all loads
    planes_request_a(0, nk, ga, ra0);
    __builtin_amdgcn_sched_barrier(0);
    planes_request_a(1, nk, ga, ra1);
    __builtin_amdgcn_sched_barrier(0);
    planes_request_b(0, nk, gb, b0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(gemm_smem + tid * 16 + i * 4096) = ra0[i];
    planes_request_a(2, nk, ga, ra0);
    __builtin_amdgcn_sched_barrier(0);
    planes_request_b(1, nk, gb, b1);
    __syncthreads();
    // (the loop's body has no branch in it: with skipped steps falling through to the loop's end the compiler sees ways
    // back to the top on which the youngest loads are the ones needed next, and waits for all loads in every step)
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
      planes_step(kt, nk, ga, gb, gemm_smem, tid, frag_a, acc, ra1, b0);
      planes_step(kt + 1, nk, ga, gb, gemm_smem, tid, frag_a, acc, ra0, b1);
    }
    if (kt < nk) planes_step(kt, nk, ga, gb, gemm_smem, tid, frag_a, acc, ra1, b0);
  }
  if (fetch_c) {
    split_epilogue<2, true>(p, acc, m0, n0, wr, wc, g, j, c_old);
  } else {
    split_epilogue<2, false>(p, acc, m0, n0, wr, wc, g, j, c_old);
  }
}

// Largest magnitudes for the two-term path, as float32 bit patterns (which order like the magnitudes; a NaN sorts
// above everything).  `diag` restricts a triangular operand to the kNb-blocks that can be non-zero: what lies in the
// blocks beyond may never have been written.
//   columns: out[c] = max(out[c], max_r |src[r][c]|), r < rows (diag: r < end of the block of c); out starts zeroed
__global__ __launch_bounds__(256) void col_absmax_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols,
                                                         int diag, unsigned* __restrict__ out) {
  __shared__ uint4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * 64 + wave * 16;
  uint4 m = make_uint4(0u, 0u, 0u, 0u);
  if (c < cols) {
    int r_end = min(r0 + 16, rows);
    if (diag) r_end = min(r_end, (c / kNb + 1) * kNb);
    for (int r = r0; r < r_end; ++r) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + static_cast<int64_t>(r) * ld + c);
      m.x = max(m.x, v.x & 0x7fffffffu);
      m.y = max(m.y, v.y & 0x7fffffffu);
      m.z = max(m.z, v.z & 0x7fffffffu);
      m.w = max(m.w, v.w & 0x7fffffffu);
    }
  }
  part[wave][lane] = m;
  __syncthreads();
  if (wave == 0 && c < cols) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const uint4 o = part[w][lane];
      m.x = max(m.x, o.x);
      m.y = max(m.y, o.y);
      m.z = max(m.z, o.z);
      m.w = max(m.w, o.w);
    }
    if (m.x) atomicMax(out + c, m.x);
    if (m.y) atomicMax(out + c + 1, m.y);
    if (m.z) atomicMax(out + c + 2, m.z);
    if (m.w) atomicMax(out + c + 3, m.w);
  }
}

//   rows: out[r] = max_c |src[r][c]|, c < cols (diag > 0: c from the start of the block of r; diag < 0: c up to the
//   end of the block of r); one wave per row
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols,
                                                         int diag, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c_begin = diag > 0 ? row / kNb * kNb : 0;
  const int c_end = diag < 0 ? min(cols, (row / kNb + 1) * kNb) : cols;
  unsigned m = 0u;
  for (int c = c_begin + 4 * lane; c < c_end; c += 256) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + static_cast<int64_t>(row) * ld + c);
    m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, static_cast<unsigned>(__shfl_xor(static_cast<int>(m), o)));
  if (lane == 0) out[row] = m;
}

// Fills absmax[0 .. M) for A and absmax[M .. M + N) for B (one vector when both operands are the same k-major matrix).
static int launch_absmax(GemmParams& p, bool a_kmajor, unsigned* absmax, hipStream_t s) {
  const bool same = a_kmajor && p.A == p.B && p.lda == p.ldb && p.M == p.N && !p.k_below_m && !p.k_below_n;
  unsigned* for_a = absmax;
  unsigned* for_b = same ? absmax : absmax + p.M;
  int rc = hip_check(hipMemsetAsync(absmax, 0, static_cast<size_t>(same ? p.N : p.M + p.N) * 4, s), "hipMemsetAsync(absmax)");
  if (rc != DCQ_OK) return rc;
  const dim3 grid_b(static_cast<unsigned>((p.N + 255) / 256), static_cast<unsigned>((p.K + 63) / 64));
  hipLaunchKernelGGL(col_absmax_kernel, grid_b, dim3(256), 0, s, p.B, p.ldb, p.K, p.N, p.k_below_n, for_b);
  DCQ_LAUNCH_CHECK("col_absmax_kernel launch");
  if (a_kmajor && !same) {
    const dim3 grid_a(static_cast<unsigned>((p.M + 255) / 256), static_cast<unsigned>((p.K + 63) / 64));
    hipLaunchKernelGGL(col_absmax_kernel, grid_a, dim3(256), 0, s, p.A, p.lda, p.K, p.M, p.k_below_m, for_a);
    DCQ_LAUNCH_CHECK("col_absmax_kernel launch");
  } else if (!a_kmajor) {
    hipLaunchKernelGGL(row_absmax_kernel, dim3(static_cast<unsigned>((p.M + 3) / 4)), dim3(256), 0, s, p.A, p.lda, p.M,
                       p.K, p.k_above_m ? 1 : (p.k_below_m ? -1 : 0), for_a);
    DCQ_LAUNCH_CHECK("row_absmax_kernel launch");
  }
  p.a_absmax = for_a;
  p.b_absmax = for_b;
  return DCQ_OK;
}

// `path`: kGemmAuto picks by problem size (results of the two float32 tile sizes are bit-identical; the split path
// rounds differently), kGemmFloat32 / kGemmSplit force one -- for callers whose results must not depend on M.
// Scratch of a two-term product: M + N words for the magnitudes, then the term planes of both operands.
static size_t two_term_planes_bytes(int64_t M, int64_t N, int64_t K) {
  return static_cast<size_t>((M + 127) / 128 + (N + 127) / 128) * static_cast<size_t>(K / kSplitK) * kImageBytes;
}
static size_t two_term_magnitude_bytes(int64_t M, int64_t N) { return static_cast<size_t>(M + N + 63) / 64 * 256; }
size_t two_term_scratch_bytes(int64_t M, int64_t N, int64_t K) {
  return two_term_magnitude_bytes(M, N) + two_term_planes_bytes(M, N, K);
}

// Magnitudes, planes and the product itself (p.vec_a / vec_b hold, N % 4 == 0, M % 4 == 0 for a k-major A).
static int launch_gemm_two_term(GemmParams& p, bool a_kmajor, unsigned char* scratch, hipStream_t s) {
  int rc = launch_absmax(p, a_kmajor, reinterpret_cast<unsigned*>(scratch), s);
  if (rc != DCQ_OK) return rc;
  const bool same = p.a_absmax == p.b_absmax;
  const int ksteps = p.K / kSplitK;
  unsigned char* planes_b = scratch + two_term_magnitude_bytes(p.M, p.N);
  unsigned char* planes_a = same ? planes_b : planes_b + static_cast<size_t>(p.tiles_n) * ksteps * kImageBytes;
  const unsigned ky = static_cast<unsigned>((ksteps + 1) / 2);
  hipLaunchKernelGGL(split_planes_kernel<true>, dim3(static_cast<unsigned>(p.tiles_n), ky), dim3(256), 0, s, p.B, p.ldb, p.N,
                     p.K, p.b_absmax, p.k_below_n ? 1 : 0, planes_b);
  DCQ_LAUNCH_CHECK("split_planes_kernel launch");
  if (a_kmajor && !same) {
    hipLaunchKernelGGL(split_planes_kernel<true>, dim3(static_cast<unsigned>(p.tiles_m), ky), dim3(256), 0, s, p.A, p.lda,
                       p.M, p.K, p.a_absmax, p.k_below_m ? 1 : 0, planes_a);
    DCQ_LAUNCH_CHECK("split_planes_kernel launch");
  } else if (!a_kmajor) {
    hipLaunchKernelGGL(split_planes_kernel<false>, dim3(static_cast<unsigned>(p.tiles_m), ky), dim3(256), 0, s, p.A, p.lda,
                       p.M, p.K, p.a_absmax, p.k_above_m ? 2 : (p.k_below_m ? 1 : 0), planes_a);
    DCQ_LAUNCH_CHECK("split_planes_kernel launch");
  }
  p.a_planes = planes_a;
  p.b_planes = planes_b;
  const unsigned grid = static_cast<unsigned>(p.tiles_m) * static_cast<unsigned>(p.tiles_n);
  hipLaunchKernelGGL(gemm_planes_kernel, dim3(grid), dim3(kGemmThreads), kPlanesLdsBytes, s, p);
  DCQ_LAUNCH_CHECK("gemm_planes_kernel launch");
  return DCQ_OK;
}

// `scratch`: NULL, or two_term_scratch_bytes(M, N, K) bytes (256-byte aligned); when given, a product that takes the
// split path runs on two float16 terms per operand (half the matrix instructions) instead of three bfloat16 terms.
static int launch_gemm(const GemmParams& pin, bool a_kmajor, hipStream_t s, int path = kGemmAuto,
                       unsigned char* scratch = nullptr) {
  GemmParams p = pin;
  if (p.M <= 0 || p.N <= 0) return DCQ_OK;
  p.vec_a = (reinterpret_cast<uintptr_t>(p.A) & 15u) == 0 && p.lda % 4 == 0;
  p.vec_b = (reinterpret_cast<uintptr_t>(p.B) & 15u) == 0 && p.ldb % 4 == 0;
  p.vec_c = (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 && p.ldc % 4 == 0;
  // fewer 128-tiles than twice the compute units: the problem is latency-bound, use 64-tiles of the float32 kernel
  const int64_t big_tiles = static_cast<int64_t>((p.M + 127) / 128) * ((p.N + 127) / 128);
  // the bf16-split kernel from 128 tiles on: alone it only pays from ~512 (n = 14336 inverse factor 28.1 -> 27.9 ms), but
  // beside other streams' kernels its shorter run per product does (112 -> 110.5 ms per Llama-3-8B layer step)
  // (K = 0: nothing to split, and no grid for the passes over K; the float32 kernel writes beta * C)
  const bool split = p.K > 0 && p.K % kSplitK == 0 && (path == kGemmSplit || (path == kGemmAuto && big_tiles >= 128));
  const int T = split || big_tiles >= 512 ? 128 : 64;
  p.tiles_m = (p.M + T - 1) / T;
  p.tiles_n = (p.N + T - 1) / T;
  // (the two-term path reads whole aligned pieces in its passes over the operands, and a k-major A clipped from
  // below has no such pass)
  if (split && scratch != nullptr && p.vec_a && p.vec_b && p.N % 4 == 0 && !(a_kmajor && (p.M % 4 != 0 || p.k_above_m)))
    return launch_gemm_two_term(p, a_kmajor, scratch, s);
  if (split) return a_kmajor ? launch_gemm_split<true>(p, s) : launch_gemm_split<false>(p, s);
  if (T == 64) return a_kmajor ? launch_gemm_variant<true, 64>(p, s) : launch_gemm_variant<false, 64>(p, s);
  return a_kmajor ? launch_gemm_variant<true, 128>(p, s) : launch_gemm_variant<false, 128>(p, s);
}

// =============================================================================================
// Diagonal block.  Cholesky elimination on the augmented block [A | I] in registers: after the 128 pivots the left
// part is R11 (upper, R11^T R11 = A) and the right part is Y = R11^-T, whose transpose X11 = R11^-1 is written into
// the inverse matrix.  One workgroup of 2 x 16 x 16 threads: waves 0-3 hold the A part, waves 4-7 the Y part, and
// thread (ty, tx) of a part owns its elements (ty + 16 i, tx + 16 c), i, c < 8 -- rows and columns dealt out
// cyclically, so that the work left shrinks for all threads alike as the pivots advance.
//
// The pivots are taken 8 at a time, with two barriers per group instead of one per pivot.  With D the group's 8 x 8
// diagonal block and D = R^T R:
//   1. the owners of the group's 8 rows put them into LDS as they stand
//   2. 256 threads, one per column of [A | Y]: each factors D for itself (R and 1 / diag R in registers) and solves
//      R^T f = (its column of the 8 rows) by forward substitution: the rows' final values                -> LDS
//   3. every row below subtracts (its entries in the 8 pivot columns) x (final rows): a rank-8 update.
// The strict lower triangle of the left part is never used (it collects garbage and is written as zero).
// info: 1-based global index of the first non-positive pivot (atomicMin: the earliest failure wins); the
// arithmetic then continues on NaNs, which is harmless (no data-dependent addressing).
// =============================================================================================
constexpr int kPanelThreads = 512;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Column t + 16 e of a part (t < 16, e < 8) is at position 8 t + e of a pivot row as it stands, so that a thread's 8
// elements of a row are two 16-byte accesses.  The final rows are kept as [part][t][pivot k][e], 72 floats per t (64
// and a pad that spreads the column writers over the banks): what thread column tx needs of all 8 pivot rows -- and,
// read at t = ty in the A part, the entries of thread rows ty + 16 i in the 8 pivot columns -- is one contiguous run.
constexpr int kPanelFinalStride = 72;
struct PanelLds {
  float raw[8][2][kNb];                      // the group's pivot rows as they stand: [row][A part | Y part][position]
  float fin[2][16][kPanelFinalStride];       // ... and their final values
};

// The two pivot groups whose rows lie in row band I0 (rows 16 I0 .. 16 I0 + 15).  Thread (part, ty, tx) owns the
// elements (ty + 16 i, tx + 16 c), i, c < 8, of its part: rows and columns are dealt out cyclically, so that what is
// left to update shrinks for every thread alike as the pivots advance -- the row bands above I0 and, in the A part,
// the column bands left of it (in the Y part: right of it) are skipped by the whole workgroup, at compile time.
// `bad`: smallest 1-based global index of a non-positive pivot seen so far.
template <int I0>
__device__ __forceinline__ void panel_groups(f32x2 (&Z)[8][4], PanelLds& lds, int32_t& bad, int32_t index_base, int tid) {
  const int tx = tid & 15, ty = (tid >> 4) & 15, part = tid >> 8;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int o = 8 * half, p0 = 16 * I0 + o;  // pivots p0 .. p0 + 7: thread rows / columns o .. o + 7 of band I0
    // 1. the pivot rows as they stand
    if (ty >= o && ty < o + 8) {
      f32x4* at = reinterpret_cast<f32x4*>(&lds.raw[ty - o][part][8 * tx]);
      at[0] = f32x4{Z[I0][0][0], Z[I0][0][1], Z[I0][1][0], Z[I0][1][1]};
      at[1] = f32x4{Z[I0][2][0], Z[I0][2][1], Z[I0][3][0], Z[I0][3][1]};
    }
    __syncthreads();
    // 2. waves 0-3 (one per SIMD), one column of [A | Y] per thread: Cholesky elimination on the 8 x 8 diagonal block
    //    D = R^T R with the thread's column x of the 8 rows carried along as a ninth column, which leaves R^-T x there
    //    (the column's entries of the final rows).  Every thread factors D for itself: the same chain of dependent
    //    operations whether one thread runs it or 256.  Each pivot's updates are applied at once to everything after
    //    it (independent operations), and nothing branches.
    if (tid < 256) {
      const int pj = tid >> 7, pos = tid & 127;
      float d[8][8], x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int c = k; c < 8; ++c) d[k][c] = lds.raw[k][0][8 * (o + c) + I0];
        x[k] = lds.raw[k][pj][pos];
      }
      float* out = &lds.fin[pj][pos >> 3][pos & 7];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = d[k][k];
        bad = min(bad, s > 0.0f ? INT32_MAX : index_base + p0 + k + 1);
        float r = __builtin_amdgcn_rsqf(s);
        r = r * (1.5f - 0.5f * s * r * r);  // one Newton step: full float32 accuracy
#pragma unroll
        for (int c = k + 1; c < 8; ++c) d[k][c] *= r;  // row k of R
        x[k] *= r;
        out[8 * k] = x[k];
#pragma unroll
        for (int i = k + 1; i < 8; ++i) {
#pragma unroll
          for (int c = i; c < 8; ++c) d[i][c] = __builtin_fmaf(-d[k][i], d[k][c], d[i][c]);
          x[i] = __builtin_fmaf(-d[k][i], x[k], x[i]);
        }
      }
    }
    __syncthreads();
    // 3. the pivot rows take their final values; every row below subtracts (its entries in the pivot columns) x
    //    (final rows).  In band I0 only the rows below the group take part (none for the band's second group).
    if (ty >= o && ty < o + 8) {
      const f32x4* at = reinterpret_cast<const f32x4*>(&lds.fin[part][tx][8 * (ty - o)]);
      const f32x4 v0 = at[0], v1 = at[1];
      Z[I0][0] = f32x2{v0[0], v0[1]};
      Z[I0][1] = f32x2{v0[2], v0[3]};
      Z[I0][2] = f32x2{v1[0], v1[1]};
      Z[I0][3] = f32x2{v1[2], v1[3]};
    }
    // column bands that can be non-zero and are still needed: A part I0 .. 7, Y part 0 .. I0 (a pair of bands per h).
    // Everything the update needs is requested from LDS before its first multiply-add.
    auto update = [&](auto lo_const, auto hi_const) __attribute__((always_inline)) {
      constexpr int lo = decltype(lo_const)::value, hi = decltype(hi_const)::value;
      const bool band_live = ty >= o + 8;  // rows of band I0 below the group
      const f32x4* mrow = reinterpret_cast<const f32x4*>(&lds.fin[0][ty][0]);     // [k][i / 4]: rows ty + 16 i
      const f32x4* crow = reinterpret_cast<const f32x4*>(&lds.fin[part][tx][0]);  // [k][c / 4]: columns tx + 16 c
      f32x4 mult[8][2], rc[8][2];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (I0 < 4) mult[k][0] = mrow[2 * k];
        mult[k][1] = mrow[2 * k + 1];
        if constexpr (lo < 2) rc[k][0] = crow[2 * k];
        if constexpr (hi > 2) rc[k][1] = crow[2 * k + 1];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        static_for<8>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i >= I0) {
            if (i > I0 || band_live) {
              const float m = -mult[k][i >> 2][i & 3];
              const f32x2 mm = {m, m};
#pragma unroll
              for (int h = lo; h < hi; ++h) {
                const f32x2 pair = {rc[k][h >> 1][2 * (h & 1)], rc[k][h >> 1][2 * (h & 1) + 1]};
                Z[i][h] = __builtin_elementwise_fma(mm, pair, Z[i][h]);
              }
            }
          }
        });
      }
    };
    if (part == 0) {
      update(std::integral_constant<int, I0 / 2>{}, std::integral_constant<int, 4>{});
    } else {
      update(std::integral_constant<int, 0>{}, std::integral_constant<int, I0 / 2 + 1>{});
    }
    // no barrier here: `raw` is rewritten before the next group's first barrier and was last read before this
    // group's second; `fin` is rewritten after the next group's first barrier, which every reader above has passed
  }
}

__device__ __forceinline__ void panel_factor(float* a, int64_t lda, float* x, int64_t ldx, int32_t* info,
                                             int32_t index_base, PanelLds& lds) {
  const int tid = threadIdx.x, tx = tid & 15, ty = (tid >> 4) & 15;
  const int part = tid >> 8;  // waves 0-3 hold the A part, waves 4-7 the Y part
  f32x2 Z[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = ty + 16 * i;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int col = tx + 16 * c;
      float v = col == r ? 1.0f : 0.0f;
      if (part == 0) {
        v = a[r * lda + col];
        v = col >= r ? v : 0.0f;
      }
      Z[i][c >> 1][c & 1] = v;
    }
  }
  int32_t bad = INT32_MAX;
  static_for<8>([&](auto i0) { panel_groups<decltype(i0)::value>(Z, lds, bad, index_base, tid); });
  if (tid == 0 && bad != INT32_MAX) atomicMin(info, bad);
  if (part == 0) {  // R11 in place (strict lower part zero)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = ty + 16 * i;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int col = tx + 16 * c;
        a[r * lda + col] = col >= r ? Z[i][c >> 1][c & 1] : 0.0f;
      }
    }
  } else {  // X11 = Y^T into the inverse matrix (upper; its lower part stays zero)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = ty + 16 * i;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int col = tx + 16 * c;  // row of X
        if (r >= col) x[col * ldx + r] = Z[i][c >> 1][c & 1];
      }
    }
  }
}

__global__ __launch_bounds__(kPanelThreads) void panel_factor_kernel(float* a, int64_t lda, float* x, int64_t ldx,
                                                                      int32_t* info, int32_t index_base) {
  __shared__ __align__(16) PanelLds lds;
  panel_factor(a, lda, x, ldx, info, index_base, lds);
}

// ---------------------------------------------------------------------------------------------
// Two adjacent diagonal blocks in ONE launch: the recursion's lowest internal node (leaf, R12, trailing update,
// leaf, X12) run by one workgroup.  Launched separately, its six kernels are a few microseconds of work each and
// spend most of their time waiting to be dispatched next to other streams' kernels; here the workgroup stays
// resident.
//
// The four 128^3 products run on the float16 matrix pipe with two terms per operand, like the large products above,
// but with everything inside the workgroup: each thread fetches its share of both operands (64 values each), the
// largest magnitudes per column (k-major operand) or row are gathered with LDS atomics, the scaled values are split
// into the four K-step images of the operand in LDS (the layout of split_planes_kernel), and the 8 waves multiply:
// wave (wr, wc, h) takes the fragments i = 2 h, 2 h + 1 of the 64 x 64 quarter (wr, wc).  The triangular operands
// are multiplied as full blocks (their zeros cost 2 us of matrix time per node, the bookkeeping would cost more).
// The images overlay the leaf's LDS: the two never live at the same time.
// ---------------------------------------------------------------------------------------------
constexpr int kPairImagesBytes = 4 * kImageBytes;  // one 128 x 128 operand: four K steps of 32
constexpr size_t kPairLdsBytes = 2 * kPairImagesBytes + 2 * kNb * sizeof(unsigned);  // images of A and B, magnitudes
static_assert(sizeof(PanelLds) <= 2 * kPairImagesBytes, "the leaf's LDS must fit under the images");
static_assert(kNb == 128 && kPanelThreads == 512, "the pair products are written for 128-blocks and 512 threads");

// A thread's 64 values of a 128 x 128 operand.  k-major (src[k][c]): columns 4 q .. 4 q + 3 of the 8 k-values of chunk
// tid / 32; row-major (src[r][k]): k = 32 (tid / 128) + 4 (u % 8) .. + 3 of the rows at image positions 16 it + u / 8.
template <bool KMAJOR>
__device__ __forceinline__ void pair_fetch(const float* src, int64_t ld, int tid, float4 (&v)[8]) {
  if constexpr (KMAJOR) {
    const int q = tid & 31, k0 = 8 * (tid >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(src + (k0 + j) * ld + 4 * q);
  } else {
    const int u = tid & 127, ks = tid >> 7;
#pragma unroll
    for (int it = 0; it < 8; ++it)
      v[it] = *reinterpret_cast<const float4*>(src + split_row_of(16 * it + (u >> 3)) * ld + 32 * ks + 4 * (u & 7));
  }
}

__device__ __forceinline__ unsigned magnitude_bits(float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; }

// amax[c] (k-major: column c; row-major: row c) = largest magnitude as a bit pattern; amax starts zeroed
template <bool KMAJOR>
__device__ __forceinline__ void pair_magnitudes(const float4 (&v)[8], int tid, unsigned* amax) {
  if constexpr (KMAJOR) {
    unsigned m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m[0] = max(m[0], magnitude_bits(v[j].x));
      m[1] = max(m[1], magnitude_bits(v[j].y));
      m[2] = max(m[2], magnitude_bits(v[j].z));
      m[3] = max(m[3], magnitude_bits(v[j].w));
    }
    const int q = tid & 31;
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicMax(&amax[4 * q + e], m[e]);
  } else {
    const int u = tid & 127;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      unsigned m = max(max(magnitude_bits(v[it].x), magnitude_bits(v[it].y)),
                       max(magnitude_bits(v[it].z), magnitude_bits(v[it].w)));
      m = max(m, static_cast<unsigned>(__shfl_xor(static_cast<int>(m), 1)));
      m = max(m, static_cast<unsigned>(__shfl_xor(static_cast<int>(m), 2)));
      m = max(m, static_cast<unsigned>(__shfl_xor(static_cast<int>(m), 4)));
      if ((u & 7) == 0) atomicMax(&amax[split_row_of(16 * it + (u >> 3))], m);
    }
  }
}

// the values, scaled into float16's range, as two terms in the operand's four images
template <bool KMAJOR>
__device__ __forceinline__ void pair_split(const float4 (&v)[8], int tid, const unsigned* amax, unsigned char* images) {
  float unused;
  if constexpr (KMAJOR) {
    const int q = tid & 31, kc = (tid >> 5) & 3, ks = tid >> 7;
    float sc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) half_scale_of(amax[4 * q + e], sc[e], unused);
    unsigned char* at = images + ks * kImageBytes + kc * kSplitChunkBytes + (((q & 15) ^ kc) + 64 * (q >> 4)) * 16;
    {
      const float x[8] = {v[0].x * sc[0], v[1].x * sc[0], v[2].x * sc[0], v[3].x * sc[0],
                          v[4].x * sc[0], v[5].x * sc[0], v[6].x * sc[0], v[7].x * sc[0]};
      split_store<2>(x, at);
    }
    {
      const float x[8] = {v[0].y * sc[1], v[1].y * sc[1], v[2].y * sc[1], v[3].y * sc[1],
                          v[4].y * sc[1], v[5].y * sc[1], v[6].y * sc[1], v[7].y * sc[1]};
      split_store<2>(x, at + 16 * 16);
    }
    {
      const float x[8] = {v[0].z * sc[2], v[1].z * sc[2], v[2].z * sc[2], v[3].z * sc[2],
                          v[4].z * sc[2], v[5].z * sc[2], v[6].z * sc[2], v[7].z * sc[2]};
      split_store<2>(x, at + 32 * 16);
    }
    {
      const float x[8] = {v[0].w * sc[3], v[1].w * sc[3], v[2].w * sc[3], v[3].w * sc[3],
                          v[4].w * sc[3], v[5].w * sc[3], v[6].w * sc[3], v[7].w * sc[3]};
      split_store<2>(x, at + 48 * 16);
    }
  } else {
    const int u = tid & 127, ks = tid >> 7, ck = (u & 7) >> 1;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int srow = 16 * it + (u >> 3);
      float scale;
      half_scale_of(amax[split_row_of(srow)], scale, unused);
      uint2 h, l;
      split_pair(f32x2{v[it].x * scale, v[it].y * scale}, h.x, l.x);
      split_pair(f32x2{v[it].z * scale, v[it].w * scale}, h.y, l.y);
      unsigned char* at = images + ks * kImageBytes + ck * kSplitChunkBytes + (srow ^ ck) * 16 + (u & 1) * 8;
      *reinterpret_cast<uint2*>(at) = h;
      *reinterpret_cast<uint2*>(at + kSplitPlaneBytes) = l;
    }
  }
}

// C = alpha (A B) + beta C from the images; lane (g, j) of wave (wr, wc, h) ends up with
// C[64 wr + 16 g + 4 r + 2 h + i][64 wc + 4 j + n], i < 2, n < 4, r < 4
__device__ __forceinline__ void pair_multiply(const unsigned char* img_a, const unsigned char* img_b, const unsigned* amax_a,
                                              const unsigned* amax_b, float alpha, float beta, float* C, int64_t ldc, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = (wave >> 1) & 1, hh = wave & 1;
  const int g = lane >> 4, j = lane & 15;
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned char* fa = img_a + g * kSplitChunkBytes + (64 * wr + (j ^ g)) * 16 + 2 * hh * 256;
  const unsigned char* fb = img_b + g * kSplitChunkBytes + (64 * wc + (j ^ g)) * 16;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 a[2][2], b[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[pl][i] = *reinterpret_cast<const u32x4*>(fa + ks * kImageBytes + pl * kSplitPlaneBytes + i * 256);
#pragma unroll
      for (int n = 0; n < 4; ++n)
        b[pl][n] = *reinterpret_cast<const u32x4*>(fb + ks * kImageBytes + pl * kSplitPlaneBytes + n * 256);
    }
    constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};  // l h', h l', h h': smallest terms first
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[pa[t]][i]),
                                                             __builtin_bit_cast(f16x8, b[pb[t]][n]), acc[i][n], 0, 0, 0);
  }
  float unused, back_b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) half_scale_of(amax_b[64 * wc + 4 * j + e], unused, back_b[e]);
  // (the values of C that are accumulated into are all requested first: read one by one between the stores, each read
  // would wait for the store before it)
  float4 old[2][4];
  if (beta != 0.0f) {  // uniform
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        old[i][r] = *reinterpret_cast<const float4*>(C + (64 * wr + 16 * g + 4 * r + 2 * hh + i) * ldc + 64 * wc + 4 * j);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 64 * wr + 16 * g + 4 * r + 2 * hh + i, col = 64 * wc + 4 * j;
      float back_a;
      half_scale_of(amax_a[row], unused, back_a);
      float4 out = make_float4(alpha * ((acc[i][0][r] * back_a) * back_b[0]), alpha * ((acc[i][1][r] * back_a) * back_b[1]),
                               alpha * ((acc[i][2][r] * back_a) * back_b[2]), alpha * ((acc[i][3][r] * back_a) * back_b[3]));
      if (beta != 0.0f)
        out = make_float4(beta * old[i][r].x + out.x, beta * old[i][r].y + out.y, beta * old[i][r].z + out.z,
                          beta * old[i][r].w + out.w);
      *reinterpret_cast<float4*>(C + row * ldc + col) = out;
    }
}

// One 128^3 product of the node.  B is k-major (B[k][n]); A is k-major (A[k][m]) or row-major; SAME: A is B, k-major
// (R12^T R12).  C may be B itself: every thread has B's values in registers before the first barrier.  On return the
// result is visible to the whole workgroup and the LDS is free.
template <bool A_KMAJOR, bool SAME>
__device__ __forceinline__ void pair_product(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                             float alpha, float beta, unsigned char* smem) {
  const int tid = threadIdx.x;
  unsigned char* img_a = smem;
  unsigned char* img_b = smem + kPairImagesBytes;
  unsigned* amax_a = reinterpret_cast<unsigned*>(smem + 2 * kPairImagesBytes);
  unsigned* amax_b = amax_a + kNb;
  float4 va[8], vb[8];
  pair_fetch<true>(B, ldb, tid, vb);
  if constexpr (!SAME) pair_fetch<A_KMAJOR>(A, lda, tid, va);
  if (tid < 2 * kNb) amax_a[tid] = 0u;
  __syncthreads();
  pair_magnitudes<true>(vb, tid, amax_b);
  if constexpr (!SAME) pair_magnitudes<A_KMAJOR>(va, tid, amax_a);
  __syncthreads();
  pair_split<true>(vb, tid, amax_b, img_b);
  if constexpr (!SAME) pair_split<A_KMAJOR>(va, tid, amax_a, img_a);
  __syncthreads();
  pair_multiply(SAME ? img_b : img_a, img_b, SAME ? amax_b : amax_a, amax_b, alpha, beta, C, ldc, tid);
  __syncthreads();
}

__global__ __launch_bounds__(kPanelThreads) void pair_factor_kernel(float* A, float* X, float* Tm, int64_t np,
                                                                     int32_t first_block, int32_t* info) {
  extern __shared__ __align__(16) unsigned char pair_smem[];  // kPairLdsBytes: the leaf's PanelLds, then the products' images
  PanelLds& panel_lds = *reinterpret_cast<PanelLds*>(pair_smem);
  const int64_t o1 = static_cast<int64_t>(first_block) * kNb, o2 = o1 + kNb;
  float* a11 = A + o1 * np + o1;
  float* a12 = A + o1 * np + o2;
  float* a22 = A + o2 * np + o2;
  float* x11 = X + o1 * np + o1;
  float* x12 = X + o1 * np + o2;
  float* x22 = X + o2 * np + o2;
  panel_factor(a11, np, x11, np, info, static_cast<int32_t>(o1), panel_lds);
  __syncthreads();
  pair_product<true, false>(x11, np, a12, np, a12, np, 1.0f, 0.0f, pair_smem);   // R12 = X11^T A12, over A12
  pair_product<true, true>(a12, np, a12, np, a22, np, -1.0f, 1.0f, pair_smem);   // A22 -= R12^T R12
  panel_factor(a22, np, x22, np, info, static_cast<int32_t>(o2), panel_lds);
  __syncthreads();
  pair_product<false, false>(a12, np, x22, np, Tm, kNb, 1.0f, 0.0f, pair_smem);  // Tm = R12 X22
  pair_product<false, false>(x11, np, Tm, kNb, x12, np, -1.0f, 0.0f, pair_smem); // X12 = -X11 Tm
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

// The same for a permuted source whose rows fit the LDS: one workgroup per destination row stages the source row
// with coalesced loads and gathers the columns from the LDS, so no global read is scattered.
constexpr int64_t kFlipRowLdsFloats = 40960;  // 160 KB
__global__ __launch_bounds__(256) void flip_pad_rows_kernel(const float* src, int64_t n, const int64_t* perm, float diag_add,
                                                            float* dst, int64_t np) {
  extern __shared__ float flip_row[];
  const int64_t i = blockIdx.x;
  float* out = dst + i * np;
  // Only the upper triangle of the destination is ever read (diagonal blocks as a whole, their lower halves masked):
  // a row is written from the first column of its diagonal block on.
  const int64_t j_first = i / kNb * kNb;
  if (i >= n) {
    for (int64_t j = j_first + threadIdx.x; j < np; j += 256) out[j] = i == j ? 1.0f : 0.0f;
    return;
  }
  const float* row = src + perm[n - 1 - i] * n;
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(perm) & 15u) == 0) {
    // 16-byte loads and stores; the four indices of destination columns j .. j + 3 are perm[n-4-j .. n-1-j], reversed
    for (int64_t j = threadIdx.x * 4; j < n; j += 256 * 4)
      *reinterpret_cast<float4*>(flip_row + j) = *reinterpret_cast<const float4*>(row + j);
    __syncthreads();
    for (int64_t j = j_first + threadIdx.x * 4; j < np; j += 256 * 4) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (j < n) {
        const longlong2 lo = *reinterpret_cast<const longlong2*>(perm + (n - 4 - j));
        const longlong2 hi = *reinterpret_cast<const longlong2*>(perm + (n - 2 - j));
        v = make_float4(flip_row[hi.y], flip_row[hi.x], flip_row[lo.y], flip_row[lo.x]);
        if (i >= j && i < j + 4) {
          float* d = i == j ? &v.x : (i == j + 1 ? &v.y : (i == j + 2 ? &v.z : &v.w));
          *d = (*d == 0.0f ? 1.0f : *d) + diag_add;
        }
      }
      *reinterpret_cast<float4*>(out + j) = v;
    }
    return;
  }
  for (int64_t j = threadIdx.x; j < n; j += 256) flip_row[j] = row[j];
  __syncthreads();
  for (int64_t j = j_first + threadIdx.x; j < np; j += 256) {
    float v = 0.0f;
    if (j < n) {
      v = flip_row[perm[n - 1 - j]];
      if (i == j) v = (v == 0.0f ? 1.0f : v) + diag_add;
    }
    out[j] = v;
  }
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
    // U is upper triangular: entries with row > column are zero and X is not read for them (its blocks strictly
    // below the diagonal hold no data)
    tile[b][tx] = (xr >= 0 && xc >= 0 && bi + tx <= bj + b) ? x[xr * np + xc] : 0.0f;  // tile[b][a]
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int64_t i = bi + a, j = bj + tx;
    if (i < n && j < n) u[i * n + j] = tile[tx][a];
  }
}

// zeroes the kNb x kNb diagonal blocks of a square matrix of leading dimension np (one workgroup per block)
__global__ __launch_bounds__(256) void zero_diagonal_blocks_kernel(float* x, int64_t np) {
  float* block = x + static_cast<int64_t>(blockIdx.x) * kNb * (np + 1);
  for (int e = threadIdx.x; e < kNb * kNb / 4; e += 256) {
    const int r = e / (kNb / 4), c4 = (e % (kNb / 4)) * 4;
    *reinterpret_cast<float4*>(block + r * np + c4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

struct CholPlan {
  int64_t np;
  int nblk;
  size_t a_bytes, x_bytes, t_bytes;
  size_t s_bytes;  // scratch of one large product on the two-term path (magnitudes and term planes); 0: none is large
  // R12 of a node stays in scratch of its own until the node is done (nothing outside the node reads it): one block
  // per recursion depth, since the second half's recursion runs while the node's R12 is still needed
  static constexpr int kMaxDepth = 24;
  size_t r_offset[kMaxDepth];  // bytes, from the start of the R12 scratch
  size_t r_bytes;
};

static CholPlan plan_chol(int64_t n) {
  CholPlan c;
  c.np = (n + kNb - 1) / kNb * kNb;
  c.nblk = static_cast<int>(c.np / kNb);
  c.a_bytes = static_cast<size_t>(c.np) * c.np * 4;
  c.x_bytes = c.a_bytes;
  const int64_t half = (c.nblk + 1) / 2 * kNb;
  c.t_bytes = static_cast<size_t>(half) * half * 4;
  // M, N and K of every product of the recursion are at most `half`; the split path starts at 128 tiles
  c.s_bytes = (half / 128) * (half / 128) >= 128 ? two_term_scratch_bytes(half, half, half) : 0;
  // a node of `size` blocks at depth d has m1 = ceil(size / 2) and m2 <= m1 blocks; sizes at most halve (rounded up)
  c.r_bytes = 0;
  int size = c.nblk;
  for (int d = 0; d < CholPlan::kMaxDepth; ++d) {
    c.r_offset[d] = c.r_bytes;
    const int64_t m1 = static_cast<int64_t>((size + 1) / 2) * kNb;
    if (size > 2) c.r_bytes += static_cast<size_t>(m1) * m1 * 4;  // (leaves and pairs keep R12 to themselves)
    size = (size + 1) / 2;
  }
  return c;
}

// Factor and invert the diagonal range [lo, hi) of blocks:  A (upper, in place) -> R with R^T R = A, and
// X = R^-1 (upper; X starts zeroed so triangular operands read zeros below their diagonal).  Recursive halves:
//     [R11, X11] = recurse(A11)
//     R12 = X11^T A12                 (triangular solve as a GEMM with the inverse that is needed anyway; kept in
//                                      scratch: only this node reads it, so A keeps A12 and no copy is needed)
//     A22 -= R12^T R12                (upper triangle only; K = size of the first half, so C is touched once)
//     [R22, X22] = recurse(A22)
//     X12 = -X11 (R12 X22)
static int factor_invert(float* A, float* X, float* T, unsigned char* mags, unsigned char* r12_scratch, const CholPlan& plan,
                         int depth, int64_t np, int lo, int hi, int32_t* info, hipStream_t s) {
  if (hi - lo == 1) {
    const int64_t off = (static_cast<int64_t>(lo) * kNb) * (np + 1);
    hipLaunchKernelGGL(panel_factor_kernel, dim3(1), dim3(kPanelThreads), 0, s, A + off, np, X + off, np, info, lo * kNb);
    DCQ_LAUNCH_CHECK("panel_factor_kernel launch");
    return DCQ_OK;
  }
  if (hi - lo == 2) {
    static std::atomic<bool> configured[64];  // (set twice by racing first calls at worst: the attribute call is idempotent)
    int dev = 0;
    int rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (rc != DCQ_OK) return rc;
    if (dev < 0 || dev >= 64 || !configured[dev]) {
      rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_factor_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPairLdsBytes)),
                     "hipFuncSetAttribute(pair LDS)");
      if (rc != DCQ_OK) return rc;
      if (dev >= 0 && dev < 64) configured[dev] = true;
    }
    hipLaunchKernelGGL(pair_factor_kernel, dim3(1), dim3(kPanelThreads), kPairLdsBytes, s, A, X, T, np, lo, info);
    DCQ_LAUNCH_CHECK("pair_factor_kernel launch");
    return DCQ_OK;
  }
  const int mid = lo + (hi - lo + 1) / 2;
  const int m1 = (mid - lo) * kNb, m2 = (hi - mid) * kNb;
  const int64_t o1 = static_cast<int64_t>(lo) * kNb, o2 = static_cast<int64_t>(mid) * kNb;
  float* a12 = A + o1 * np + o2;
  int rc = factor_invert(A, X, T, mags, r12_scratch, plan, depth + 1, np, lo, mid, info, s);
  if (rc != DCQ_OK) return rc;
  // R12 [m1][m2] lives in this depth's scratch block: nothing outside the node reads it (its place in A keeps A12),
  // and it must outlast the recursion on the second half, which uses T and the deeper blocks
  DCQ_REQUIRE(depth < CholPlan::kMaxDepth, DCQ_ERR_UNSUPPORTED, "recursion deeper than planned");
  float* r12 = reinterpret_cast<float*>(r12_scratch + plan.r_offset[depth]);
  {  // R12 = X11^T A12
    GemmParams g{};
    g.A = X + o1 * np + o1;  // k-major operand: A'(m, k) = X11[k][m], zero for k > m
    g.lda = np;
    g.B = a12;
    g.ldb = np;
    g.C = r12;
    g.ldc = m2;
    g.M = m1;
    g.N = m2;
    g.K = m1;
    g.alpha = 1.0f;
    g.beta = 0.0f;
    g.k_below_m = 1;
    rc = launch_gemm(g, true, s, kGemmAuto, mags);
    if (rc != DCQ_OK) return rc;
  }
  {  // A22 -= R12^T R12 (both operands are the rows of R12)
    GemmParams g{};
    g.A = r12;
    g.lda = m2;
    g.B = r12;
    g.ldb = m2;
    g.C = A + o2 * np + o2;
    g.ldc = np;
    g.M = m2;
    g.N = m2;
    g.K = m1;
    g.alpha = -1.0f;
    g.beta = 1.0f;
    g.upper_only = 1;
    rc = launch_gemm(g, true, s, kGemmAuto, mags);
    if (rc != DCQ_OK) return rc;
  }
  rc = factor_invert(A, X, T, mags, r12_scratch, plan, depth + 1, np, mid, hi, info, s);
  if (rc != DCQ_OK) return rc;
  {  // T = R12 X22 (X22 upper: k <= n)
    GemmParams g{};
    g.A = r12;
    g.lda = m2;
    g.B = X + o2 * np + o2;
    g.ldb = np;
    g.C = T;
    g.ldc = m2;
    g.M = m1;
    g.N = m2;
    g.K = m2;
    g.alpha = 1.0f;
    g.beta = 0.0f;
    g.k_below_n = 1;
    rc = launch_gemm(g, false, s, kGemmAuto, mags);
    if (rc != DCQ_OK) return rc;
  }
  {  // X12 = -X11 T (X11 upper: k >= m)
    GemmParams g{};
    g.A = X + o1 * np + o1;
    g.lda = np;
    g.B = T;
    g.ldb = m2;
    g.C = X + o1 * np + o2;
    g.ldc = np;
    g.M = m1;
    g.N = m2;
    g.K = m1;
    g.alpha = -1.0f;
    g.beta = 0.0f;
    g.k_above_m = 1;
    rc = launch_gemm(g, false, s, kGemmAuto, mags);
    if (rc != DCQ_OK) return rc;
  }
  return DCQ_OK;
}

int cholesky_inverse_upper_on(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U, void* workspace,
                              int32_t* info, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Built-in scratch for dcq_cholesky_inverse_upper_static: callers that may not allocate (HIP graph capture, serving
// loops) get the factorisation's workspace from this file-scope device array instead of an argument.  Every GPU the
// library runs on holds its own copy at its own address; `scratch` itself is only the host-side handle of the
// symbol, so the address is looked up once per device (hipGetSymbolAddress with that device current) and only that
// pointer is ever handed to kernels and memsets.
// ---------------------------------------------------------------------------------------------
constexpr int kStaticCholMaxN = 2048;
// matrix, inverse, the product scratch of half the order, and the R12 blocks of the three recursion depths above a pair
constexpr size_t kStaticCholBytes = 2ull * kStaticCholMaxN * kStaticCholMaxN * 4 + 1ull * (kStaticCholMaxN / 2) * (kStaticCholMaxN / 2) * 4 +
                                    (1ull * (kStaticCholMaxN / 2) * (kStaticCholMaxN / 2) + 1ull * (kStaticCholMaxN / 4) * (kStaticCholMaxN / 4) +
                                     1ull * (kStaticCholMaxN / 8) * (kStaticCholMaxN / 8)) * 4;
static __device__ __align__(256) unsigned char scratch[kStaticCholBytes];  // 41.25 MiB per GPU, for the process's lifetime

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

static int static_scratch_of_current_device(void** out) {
  int dev = 0;
  int rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
  if (rc != DCQ_OK) return rc;
  DCQ_REQUIRE(dev >= 0 && dev < kMaxDevices, DCQ_ERR_UNSUPPORTED, "device index %d out of range", dev);
  std::lock_guard<std::mutex> guard(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* ptr = nullptr;
    rc = hip_check(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
    if (rc != DCQ_OK) return rc;
    DCQ_REQUIRE(ptr != nullptr && (reinterpret_cast<uintptr_t>(ptr) & 255u) == 0, DCQ_ERR_HIP,
                "scratch symbol resolved to a bad address");
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return DCQ_OK;
}

}  // namespace dcq

using namespace dcq;

namespace dcq {

int gemm_f32_on_path(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, float alpha, float beta, hipStream_t stream, int path,
                     unsigned char* scratch) {
  DCQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (M == 0 || N == 0) return DCQ_OK;
  DCQ_REQUIRE(A && B && C, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(K % 32 == 0, DCQ_ERR_UNSUPPORTED, "K must be a multiple of 32 (got %lld)", (long long)K);
  DCQ_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), DCQ_ERR_UNSUPPORTED, "size too large");
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
  return launch_gemm(g, a_kmajor != 0, stream, path, scratch);
}

}  // namespace dcq

extern "C" {

int dcq_gemm_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, float alpha, float beta, dcq_stream_t stream) {
  return gemm_f32_on_path(A, lda, a_kmajor, B, ldb, C, ldc, M, N, K, alpha, beta, static_cast<hipStream_t>(stream), kGemmAuto, nullptr);
}

int64_t dcq_gemm_f32_two_term_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return static_cast<int64_t>(two_term_scratch_bytes(M, N, K));
}

int dcq_gemm_f32_two_term(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, float alpha, float beta, void* workspace, int64_t workspace_bytes,
                          dcq_stream_t stream) {
  DCQ_REQUIRE(M >= 0 && N >= 0 && K >= 0, DCQ_ERR_ARGUMENT, "negative size");
  if (M == 0 || N == 0) return DCQ_OK;
  DCQ_REQUIRE(A && B && C && workspace, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(K % 32 == 0 && N % 4 == 0 && (!a_kmajor || M % 4 == 0), DCQ_ERR_UNSUPPORTED,
              "K must be a multiple of 32, N (and M of a k-major A) of 4 (got M %lld N %lld K %lld)", (long long)M,
              (long long)N, (long long)K);
  DCQ_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), DCQ_ERR_UNSUPPORTED, "size too large");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(B) & 15u) == 0 && lda % 4 == 0 &&
                  ldb % 4 == 0,
              DCQ_ERR_UNSUPPORTED, "A and B must be 16-byte aligned with leading dimensions that are multiples of 4");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DCQ_ERR_ARGUMENT, "workspace must be 256-byte aligned");
  DCQ_REQUIRE(workspace_bytes >= dcq_gemm_f32_two_term_workspace_bytes(M, N, K), DCQ_ERR_ARGUMENT,
              "workspace too small: dcq_gemm_f32_two_term_workspace_bytes(M, N, K) bytes are needed");
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
  return launch_gemm(g, a_kmajor != 0, static_cast<hipStream_t>(stream), kGemmSplit, static_cast<unsigned char*>(workspace));
}

int64_t dcq_cholesky_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const CholPlan c = plan_chol(n);
  return static_cast<int64_t>(c.a_bytes + c.x_bytes + c.t_bytes + c.s_bytes + c.r_bytes);
}

int dcq_cholesky_inverse_upper(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U, void* workspace,
                               int64_t workspace_bytes, int32_t* info, dcq_stream_t stream) {
  DCQ_REQUIRE(n > 0 && n < (1 << 20), DCQ_ERR_ARGUMENT, "bad n");
  DCQ_REQUIRE(H && U && workspace && info, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DCQ_ERR_ARGUMENT, "workspace must be 256-byte aligned");
  DCQ_REQUIRE(workspace_bytes >= dcq_cholesky_workspace_bytes(n), DCQ_ERR_ARGUMENT, "workspace too small");
  return cholesky_inverse_upper_on(H, n, perm, diag_add, U, workspace, info, static_cast<hipStream_t>(stream));
}

int dcq_cholesky_inverse_upper_static(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U,
                                      int32_t* info, dcq_stream_t stream) {
  DCQ_REQUIRE(n > 0, DCQ_ERR_ARGUMENT, "bad n");
  DCQ_REQUIRE(n <= kStaticCholMaxN, DCQ_ERR_UNSUPPORTED,
              "n = %lld exceeds the built-in scratch of dcq_cholesky_inverse_upper_static (n <= %d); use "
              "dcq_cholesky_inverse_upper with a caller-provided workspace",
              (long long)n, kStaticCholMaxN);
  DCQ_REQUIRE(H && U && info, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(static_cast<size_t>(dcq_cholesky_workspace_bytes(n)) <= kStaticCholBytes, DCQ_ERR_UNSUPPORTED,
              "the built-in scratch is smaller than the workspace n = %lld needs", (long long)n);
  void* ws = nullptr;
  const int rc = static_scratch_of_current_device(&ws);
  if (rc != DCQ_OK) return rc;
  return cholesky_inverse_upper_on(H, n, perm, diag_add, U, ws, info, static_cast<hipStream_t>(stream));
}

}  // extern "C"

namespace dcq {

int cholesky_inverse_upper_on(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U, void* workspace,
                              int32_t* info, hipStream_t s) {
  const CholPlan c = plan_chol(n);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  float* A = reinterpret_cast<float*>(ws);
  float* X = reinterpret_cast<float*>(ws + c.a_bytes);
  float* T = reinterpret_cast<float*>(ws + c.a_bytes + c.x_bytes);
  unsigned char* mags = c.s_bytes ? ws + c.a_bytes + c.x_bytes + c.t_bytes : nullptr;
  const int64_t np = c.np;
  int rc = hip_check(hipMemsetAsync(info, 0x7f, 4, s), "hipMemsetAsync(info)");  // 0x7f7f7f7f = "no failure yet"
  if (rc != DCQ_OK) return rc;
  if (perm != nullptr && n <= kFlipRowLdsFloats) {
    const size_t lds = static_cast<size_t>(n) * sizeof(float);
    if (lds > 64 * 1024) {
      rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(flip_pad_rows_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFlipRowLdsFloats * 4)),
                     "hipFuncSetAttribute(flip_pad_rows_kernel)");
      if (rc != DCQ_OK) return rc;
    }
    hipLaunchKernelGGL(flip_pad_rows_kernel, dim3(static_cast<unsigned>(np)), dim3(256), lds, s, H, n, perm, diag_add, A, np);
    DCQ_LAUNCH_CHECK("flip_pad_rows_kernel launch");
  } else {
    hipLaunchKernelGGL(flip_pad_kernel, dim3(static_cast<unsigned>((np + 255) / 256), static_cast<unsigned>(np)),
                       dim3(256), 0, s, H, n, perm, diag_add, A, np);
    DCQ_LAUNCH_CHECK("flip_pad_kernel launch");
  }
  // The inverse is assembled in its upper triangle; triangular operands are read in whole kNb-blocks, so the lower
  // halves of the diagonal blocks must read as zero.  Nothing ever reads a block strictly below the diagonal.
  hipLaunchKernelGGL(zero_diagonal_blocks_kernel, dim3(static_cast<unsigned>(c.nblk)), dim3(256), 0, s, X, np);
  DCQ_LAUNCH_CHECK("zero_diagonal_blocks_kernel launch");
  rc = factor_invert(A, X, T, mags, ws + c.a_bytes + c.x_bytes + c.t_bytes + c.s_bytes, c, 0, np, 0, c.nblk, info, s);
  if (rc != DCQ_OK) return rc;
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(static_cast<unsigned>((n + 31) / 32), static_cast<unsigned>((n + 31) / 32)),
                     dim3(256), 0, s, X, np, U, n);
  DCQ_LAUNCH_CHECK("flip_transpose_kernel launch");
  hipLaunchKernelGGL(finish_info_kernel, dim3(1), dim3(64), 0, s, info);
  DCQ_LAUNCH_CHECK("finish_info_kernel launch");
  return DCQ_OK;
}

}  // namespace dcq
