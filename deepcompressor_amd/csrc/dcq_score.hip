// Smoothing candidates of one linear module scored on its outputs (gfx950).
//
// The reference scores a candidate scale s of a linear module on that module's outputs (OutputsError objective,
// deepcompressor/calib/search.py:1011-1042 with the candidate put in place by calib/smooth.py:575-625):
//     y = (Q(x / s) * s) @ (Q(w * s) / s)^T + b,   error = sum (y - y0)^2
// With integer quantizers both operands are (integer code) x (scale of the code's group) x s^(+-1), and the s of a
// channel cancels in every product: y[t][r] = sum_g sx[t][g] sw[r][g] sum_{c in g} cx[t][c] cw[r][c].  The inner sums
// are exact in int32 on the int8 matrix cores, so a candidate costs one pass that writes codes (dcq_smooth_quantize_groups)
// and this GEMM whose epilogue takes the difference from the target outputs and sums the squares -- no processed
// tensor, no output tensor.  The reference forms the same products in float32 after rounding q * scale and the
// multiplication / division by s (a few units of 2^-24 per term); the errors agree to that rounding.
#include "dcq_common.cuh"

namespace dcq {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

constexpr int kScoreTile = 128;                   // tokens and output rows of one workgroup's tile
constexpr int kScoreStage = 128;                  // bytes of the reduction staged through LDS at a time (narrow tiles)
constexpr int kScoreThreads = 256;

struct ScoreParams {
  const int8_t* xc;   // [K][T][C]
  const float* xs;    // [K][T][x_groups]
  const int8_t* wc;   // [K][R][C]
  const float* ws;    // [K][R][w_groups]
  int64_t T, R, C;
  int32_t x_groups, w_groups;  // per row
  int32_t fold_stages;         // LDS stages between two foldings of the int32 sums into the float32 outputs
  int32_t x_folds, w_folds;    // foldings per group of the tokens / of the rows
  int32_t tiles_t, tiles_r;
  int64_t blocks, blocks_per_xcd;  // workgroups with work, and ceil(blocks / 8)
  const float* bias;           // [R] or NULL
  const float* target;         // [T][ld_target]
  int64_t ld_target;
  double* partial;             // [K][tiles_t * tiles_r]
};

__device__ __forceinline__ int score_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// One workgroup: a tile of 128 tokens x (64 WN) output rows of one candidate.  Four waves, 2 x 2, each 64 tokens x
// (32 WN) rows as 2 x WN matrix-core tiles of 32 x 32 (K = 32 bytes per instruction); both operands are rows of
// codes, so a lane's fragment is 16 consecutive bytes of a row for x and for w alike.  Global -> registers -> LDS,
// double buffered, one barrier per stage, two workgroups per CU.  (WN = 2 is the one launched: 256-row tiles, WN = 4,
// halve the bytes read per product but spill registers in the epilogue and ran slower.)
template <bool MULTI, int WN>
__global__ __launch_bounds__(kScoreThreads, MULTI ? 1 : 2) void score_kernel(const ScoreParams p) {
  constexpr int kTileR = 64 * WN;
  constexpr int kStage = WN > 2 ? kScoreStage / 2 : kScoreStage;  // two workgroups per CU either way
  constexpr int kScoreStride = kStage + 16;  // LDS row pitch: 16 lanes' 16-byte reads land on 64 distinct banks
  constexpr int kPieces = kStage / 16;       // 16-byte pieces of a row per stage
  constexpr int kRowsPerPass = kScoreThreads / kPieces;
  constexpr int kScoreChunks = kScoreTile / kRowsPerPass, kChunksW = kTileR / kRowsPerPass;
  __shared__ __attribute__((aligned(16))) int8_t lds_x[2][kScoreTile * kScoreStride];
  __shared__ __attribute__((aligned(16))) int8_t lds_w[2][kTileR * kScoreStride];
  __shared__ double wave_sum[kScoreThreads / kWave];
  // Workgroups go to the 8 XCDs round-robin, and each XCD has its own L2: XCD x takes the x-th contiguous eighth of
  // the (candidate, row tile, token tile) order, so the workgroups resident on it at one time share a few row tiles
  // and the token tiles of one candidate, which its L2 holds.
  const int tiles = p.tiles_t * p.tiles_r;
  const int64_t n = static_cast<int64_t>(blockIdx.x & 7) * p.blocks_per_xcd + (blockIdx.x >> 3);
  if (n >= p.blocks) return;
  const int k = static_cast<int>(n / tiles);
  const int tile = static_cast<int>(n - static_cast<int64_t>(k) * tiles);
  const int tile_r = tile / p.tiles_t, tile_t = tile - tile_r * p.tiles_t;
  const int64_t t0 = static_cast<int64_t>(tile_t) * kScoreTile, r0 = static_cast<int64_t>(tile_r) * kTileR;
  const int8_t* xc = p.xc + static_cast<int64_t>(k) * p.T * p.C;
  const int8_t* wc = p.wc + static_cast<int64_t>(k) * p.R * p.C;
  const float* xs = p.xs + static_cast<int64_t>(k) * p.T * p.x_groups;
  const float* ws = p.ws + static_cast<int64_t>(k) * p.R * p.w_groups;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, half = lane >> 5;

  // this thread's pieces of a stage: piece q -> row q / 8 of the tile, bytes 16 (q % 8) .. of the stage
  i32x4 stage_x[kScoreChunks], stage_w[kChunksW];
  // (loads outside the tensors are redirected to a valid address and their result replaced by zeros: no branches, so
  // the loads of a stage are issued together)
  auto fetch = [&](int64_t c0) {
    const i32x4 zero = {0, 0, 0, 0};
    const int64_t c = c0 + (threadIdx.x % kPieces) * 16;
    const bool in_c = c < p.C;  // C is a multiple of 16: a piece is inside or outside as a whole
    const int64_t cc = in_c ? c : 0;
#pragma unroll
    for (int i = 0; i < kScoreChunks; ++i) {
      const int64_t row = t0 + threadIdx.x / kPieces + i * kRowsPerPass;
      const i32x4 v = *reinterpret_cast<const i32x4*>(xc + min(row, p.T - 1) * p.C + cc);
      stage_x[i] = (in_c && row < p.T) ? v : zero;
    }
#pragma unroll
    for (int i = 0; i < kChunksW; ++i) {
      const int64_t row = r0 + threadIdx.x / kPieces + i * kRowsPerPass;
      const i32x4 v = *reinterpret_cast<const i32x4*>(wc + min(row, p.R - 1) * p.C + cc);
      stage_w[i] = (in_c && row < p.R) ? v : zero;
    }
  };
  auto stash = [&](int buffer) {
    const int at = (threadIdx.x / kPieces) * kScoreStride + (threadIdx.x % kPieces) * 16;
#pragma unroll
    for (int i = 0; i < kScoreChunks; ++i)
      *reinterpret_cast<i32x4*>(&lds_x[buffer][at + i * kRowsPerPass * kScoreStride]) = stage_x[i];
#pragma unroll
    for (int i = 0; i < kChunksW; ++i)
      *reinterpret_cast<i32x4*>(&lds_w[buffer][at + i * kRowsPerPass * kScoreStride]) = stage_w[i];
  };

  constexpr int YI = MULTI ? 2 : 1, YJ = MULTI ? WN : 1;
  i32x16 acc[2][WN];
  float y[YI][YJ][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
#pragma unroll
  for (int i = 0; i < YI; ++i)
#pragma unroll
    for (int j = 0; j < YJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) y[i][j][e] = 0.0f;

  // int32 sums -> float32, times the scales of group gx of the tokens and group gw of the rows (rows and tokens
  // outside the tensors hold zero codes, so any scale does for them)
  auto fold = [&](int gx, int gw, float (&out)[16], const i32x16& sums, int i, int j, bool add) {
    const int64_t r = min(r0 + wn * (32 * WN) + j * 32 + lr, p.R - 1);
    const float sw = ws[r * p.w_groups + gw];
    float sx[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sx[e] = xs[min(t0 + wm * 64 + i * 32 + score_row(e, half), p.T - 1) * p.x_groups + gx];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = static_cast<float>(sums[e]) * sx[e] * sw;
      out[e] = add ? out[e] + v : v;
    }
  };

  const int stages = static_cast<int>((p.C + kStage - 1) / kStage);
  fetch(0);
  stash(0);
  __syncthreads();
  for (int s = 0; s < stages; ++s) {
    const int cur = s & 1;
    if (s + 1 < stages) fetch(static_cast<int64_t>(s + 1) * kStage);
    const int8_t* ax = &lds_x[cur][(wm * 64 + lr) * kScoreStride + half * 16];
    const int8_t* bw = &lds_w[cur][(wn * (32 * WN) + lr) * kScoreStride + half * 16];
#pragma unroll
    for (int ks = 0; ks < kStage / 32; ++ks) {
      i32x4 a[2], b[WN];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const i32x4*>(ax + i * 32 * kScoreStride + ks * 32);
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const i32x4*>(bw + j * 32 * kScoreStride + ks * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (MULTI) {
      // (a fold is a whole number of stages, the channels a whole number of folds)
      if ((s + 1) % p.fold_stages == 0) {
        const int f = s / p.fold_stages;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            fold(f / p.x_folds, f / p.w_folds, y[MULTI ? i : 0][MULTI ? j : 0], acc[i][j], i, j, true);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
          }
      }
    }
    if (s + 1 < stages) {
      stash(cur ^ 1);  // (the buffer read two stages ago: every wave passed the barrier that followed those reads)
      __syncthreads();
    }
  }

  // ---- epilogue: (y + b - y0)^2 over the tile
  float err = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float single[16];
      if (!MULTI) fold(0, 0, single, acc[i][j], i, j, false);
      const float(&v)[16] = MULTI ? y[MULTI ? i : 0][MULTI ? j : 0] : single;
      const int64_t r = r0 + wn * (32 * WN) + j * 32 + lr;
      const int64_t rc = min(r, p.R - 1);
      const float b = p.bias != nullptr ? p.bias[rc] : 0.0f;
      float want[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        want[e] = p.target[min(t0 + wm * 64 + i * 32 + score_row(e, half), p.T - 1) * p.ld_target + rc];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t t = t0 + wm * 64 + i * 32 + score_row(e, half);
        const float d = (t < p.T && r < p.R) ? (v[e] + b) - want[e] : 0.0f;
        err = err + d * d;
      }
    }
  double total = static_cast<double>(err);
  for (int w = kWave >> 1; w > 0; w >>= 1) total += __shfl_xor(total, w, kWave);
  if (lane == 0) wave_sum[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;
    for (int w = 0; w < kScoreThreads / kWave; ++w) sum += wave_sum[w];
    p.partial[static_cast<int64_t>(k) * tiles + tile] = sum;
  }
}

// errors[k] (+)= sum of the candidate's tile sums, in a fixed order
__global__ __launch_bounds__(256) void score_reduce_kernel(const double* partial, int tiles, double* errors, int accumulate) {
  __shared__ double part[256];
  const double* row = partial + static_cast<int64_t>(blockIdx.x) * tiles;
  double sum = 0.0;
  for (int i = threadIdx.x; i < tiles; i += 256) sum += row[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) errors[blockIdx.x] = accumulate ? errors[blockIdx.x] + part[0] : part[0];
}

}  // namespace dcq

extern "C" int64_t dcq_linear_candidate_errors_workspace(int64_t T, int64_t R, int32_t K) {
  if (T <= 0 || R <= 0 || K <= 0) return 0;
  const int64_t tiles = ((T + dcq::kScoreTile - 1) / dcq::kScoreTile) * ((R + dcq::kScoreTile - 1) / dcq::kScoreTile);
  return tiles * K * static_cast<int64_t>(sizeof(double));
}

extern "C" int dcq_linear_candidate_errors(const int8_t* x_codes, const float* x_scales, int64_t T, int64_t x_group,
                                           const int8_t* w_codes, const float* w_scales, int64_t R, int64_t w_group,
                                           int64_t C, int32_t K, const float* bias, const float* target,
                                           int64_t ld_target, void* workspace, int64_t workspace_bytes, double* errors,
                                           int32_t accumulate, dcq_stream_t stream) {
  using namespace dcq;
  DCQ_REQUIRE(T >= 0 && R >= 0 && C > 0 && K > 0, DCQ_ERR_ARGUMENT, "bad sizes");
  DCQ_REQUIRE(errors != nullptr, DCQ_ERR_ARGUMENT, "errors is NULL");
  DCQ_REQUIRE(x_group > 0 && w_group > 0 && C % x_group == 0 && C % w_group == 0, DCQ_ERR_ARGUMENT,
              "the channels must be whole groups");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (T == 0 || R == 0) {
    if (accumulate) return DCQ_OK;
    return hip_check(hipMemsetAsync(errors, 0, sizeof(double) * K, s), "hipMemsetAsync(errors)");
  }
  DCQ_REQUIRE(x_codes && x_scales && w_codes && w_scales && target, DCQ_ERR_ARGUMENT, "NULL buffer");
  auto aligned = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) % 16) == 0; };
  DCQ_REQUIRE(C % 16 == 0 && aligned(x_codes) && aligned(w_codes), DCQ_ERR_UNSUPPORTED,
              "rows of codes must be whole 16-byte pieces of 16-byte aligned buffers");
  DCQ_REQUIRE(ld_target >= R, DCQ_ERR_ARGUMENT, "ld_target is shorter than a row of outputs");
  const int64_t small = x_group < w_group ? x_group : w_group, large = x_group < w_group ? w_group : x_group;
  const bool multi = small < C;
  // |code x code| <= 2^14: an int32 sum holds 2^17 of them
  DCQ_REQUIRE(small <= (1 << 17), DCQ_ERR_UNSUPPORTED, "more than 131072 channels between two foldings of the integer sums");
  DCQ_REQUIRE(!multi || (large % small == 0 && small % kScoreStage == 0), DCQ_ERR_UNSUPPORTED,
              "groups inside a row must be multiples of 128 channels, one size a multiple of the other");
  const int64_t tile_rows = kScoreTile;
  const int64_t tiles_t = (T + kScoreTile - 1) / kScoreTile, tiles_r = (R + tile_rows - 1) / tile_rows;
  const int64_t blocks = tiles_t * tiles_r * K;
  DCQ_REQUIRE(blocks < (int64_t{1} << 31) - 8 && tiles_t * tiles_r < (1 << 24), DCQ_ERR_UNSUPPORTED, "too many tiles");
  DCQ_REQUIRE(workspace != nullptr && workspace_bytes >= dcq_linear_candidate_errors_workspace(T, R, K), DCQ_ERR_ARGUMENT,
              "workspace is smaller than dcq_linear_candidate_errors_workspace says");
  ScoreParams p{};
  p.xc = x_codes, p.xs = x_scales, p.wc = w_codes, p.ws = w_scales;
  p.T = T, p.R = R, p.C = C;
  p.x_groups = static_cast<int32_t>(C / x_group), p.w_groups = static_cast<int32_t>(C / w_group);
  p.fold_stages = static_cast<int32_t>(multi ? small / kScoreStage : 1);
  p.x_folds = static_cast<int32_t>(x_group / small), p.w_folds = static_cast<int32_t>(w_group / small);
  p.tiles_t = static_cast<int32_t>(tiles_t), p.tiles_r = static_cast<int32_t>(tiles_r);
  p.blocks = blocks, p.blocks_per_xcd = (blocks + 7) / 8;
  const int64_t grid = p.blocks_per_xcd * 8;
  p.bias = bias, p.target = target, p.ld_target = ld_target;
  p.partial = static_cast<double*>(workspace);
  if (multi) {
    hipLaunchKernelGGL((score_kernel<true, 2>), dim3(static_cast<unsigned>(grid)), dim3(kScoreThreads), 0, s, p);
  } else {
    hipLaunchKernelGGL((score_kernel<false, 2>), dim3(static_cast<unsigned>(grid)), dim3(kScoreThreads), 0, s, p);
  }
  DCQ_LAUNCH_CHECK("score_kernel launch");
  hipLaunchKernelGGL(score_reduce_kernel, dim3(static_cast<unsigned>(K)), dim3(256), 0, s, p.partial,
                     static_cast<int>(tiles_t * tiles_r), errors, accumulate);
  DCQ_LAUNCH_CHECK("score_reduce_kernel launch");
  return DCQ_OK;
}
