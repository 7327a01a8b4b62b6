// GPTQ kernels (gfx950): Hessian accumulation H += alpha * X^T X on the matrix cores, and the
// column sweep of one 128-column block (quantize, propagate the error inside the block).
//
// Hessian.  X is [T][C] 16-bit, row-major (tokens x channels).  Both MFMA operands of X^T X need 8
// consecutive *tokens* of one channel per lane, so X is first transposed once into a zero-padded
// Xt[Cp][Tp] (a pure copy, ~2 passes over X against ~C/64 passes of arithmetic); the contraction is
// then the standard "A [M][K] times B [N][K] transposed" form with both fragments contiguous 16-byte
// LDS reads.  Products of 16-bit inputs are exact in float32, accumulation is float32.  Only tiles
// on or above the diagonal are computed and mirrored.  K (=tokens) is split over several workgroups
// when the tile count alone cannot fill the chip; partial tiles go to a workspace slab and are
// summed in a fixed order, so the result is bitwise reproducible.
//
// Sweep.  Rows of W are independent in GPTQ.  A team of 16 lanes owns one row of the 128-column
// block (8 columns per lane, in registers); the upper-triangular 128x128 block of U (U^T U = H^-1)
// sits in LDS.  For each column the owner lane quantizes, the error is broadcast inside the team and
// every lane updates its remaining columns.
#include "dcq_common.cuh"

#include <type_traits>

namespace dcq {

// =============================================================================================
// transpose + pad
// =============================================================================================
constexpr int kTrTile = 64;

__global__ __launch_bounds__(256) void transpose_pad_kernel(const uint16_t* __restrict__ x, int64_t T, int64_t C,
                                                            uint16_t* __restrict__ xt, int64_t ktiles) {
  __shared__ uint16_t tile[kTrTile][kTrTile + 2];
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * kTrTile;
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * kTrTile;
  // load: 64 rows (tokens) x 64 channels, 8 channels (16 B) per thread, 32 rows per pass
  const int lr = threadIdx.x / 8, lc = (threadIdx.x % 8) * 8;
  const bool vec_ok = (C % 8) == 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = lr + pass * 32;
    const int64_t t = t0 + r;
    uint16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < T) {
      if (vec_ok && c0 + lc + 8 <= C) {
        const uint4 raw = *reinterpret_cast<const uint4*>(x + t * C + c0 + lc);
        const uint16_t* pr = reinterpret_cast<const uint16_t*>(&raw);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pr[i];
      } else {
        for (int i = 0; i < 8; ++i)
          if (c0 + lc + i < C) v[i] = x[t * C + c0 + lc + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[r][lc + i] = v[i];
  }
  __syncthreads();
  // store: 64 channels x 64 tokens, 8 tokens (16 B) per thread
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = lr + pass * 32;
    uint4 raw;
    uint16_t* pr = reinterpret_cast<uint16_t*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) pr[i] = tile[lc + i][c];
    // blocked layout [panel of 256 channels][k-tile of 64 tokens][channel in panel][token in k-tile]: the 32 KiB
    // a SYRK workgroup stages per operand and k-tile are one contiguous block
    const int64_t ch = c0 + c;
    *reinterpret_cast<uint4*>(xt + (((ch >> 8) * ktiles + blockIdx.x) * 256 + (ch & 255)) * 64 + lc) = raw;
  }
}

// =============================================================================================
// SYRK on MFMA: tile 256 x 256, K step 64, 8 waves (2 x 4 grid of 128 x 64 sub-tiles), 16x16x32 instructions (the
// chip holds a higher clock on them than on the 32x32x16 form: +5 % measured here), fragment reads issued one
// group of 4 instructions ahead of their use (+5 % more).
// A 256-tile needs 128 flop per operand byte fetched from L2, twice what a 128-tile does: at the matrix-core peak
// the chip-wide L2 -> LDS stream of 128-tiles would be ~39 TB/s, about twice what the L2s deliver.
//
// Operand tiles go HBM/L2 -> LDS directly (global_load_lds, 16 B per lane, no VGPR staging).  The LDS image
// of a tile is row-major [256 rows][8 chunks of 16 B]; one wave-instruction fills 8 consecutive rows, its 8
// lanes per row fetch one full 128-byte line.  To keep the 16-byte fragment reads conflict-free the chunk
// position is XOR-swizzled with bits 1..3 of the row; direct-to-LDS writes are lane-linear, so the swizzle is
// applied to the *global source* chunk and again on the read.  In memory the transposed activations are blocked
// the same way, [panel of 256 channels][k-tile][256 rows][64 tokens]: what a workgroup stages per operand and
// k-tile is one contiguous 32 KiB block, every wave-instruction of the staging reads 1 KiB of consecutive bytes
// (rows of a plain [channel][token] matrix are half a megabyte apart: 512 separate 128-byte lines per k-tile).
//
// Workgroups are renumbered so that the workgroups resident on one XCD (blockIdx = xcd mod 8) walk one
// 8 x 8 patch of output tiles: every row/column panel streamed from memory is then shared by several workgroups
// through that XCD's L2.
// =============================================================================================
constexpr int kTile = 256;
constexpr int kBK = 64;
constexpr int kRowBytes = kBK * 2;                // 128 B per tile row
constexpr int kOperandBytes = kTile * kRowBytes;  // 32 KiB
constexpr int kSyrkThreads = 512;
constexpr int kPatch = 8;
constexpr int kDirectMaxC = 8192;  // widest activations the SYRK reads untransposed

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ST>
__device__ __forceinline__ f32x4 mfma_16x16x32(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (ST == DCQ_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
}

// Direct-to-LDS load of 16 B per lane: the wave's 64 pieces land lane-linearly at `lds_wave_base` (M0).  Issued as
// inline assembly on purpose: through the builtin the compiler treats the LDS write as aliasing every later LDS
// read and drains the load (vmcnt(0)) before the first fragment read, which serialises the fetch of k-tile
// kt + 1 with the arithmetic of k-tile kt.  The caller owns the ordering instead: wait_direct_loads() + barrier
// before the filled buffer is read, and never a read of the buffer being filled.
__device__ __forceinline__ void load_to_lds_16(const void* gsrc, void* lds_wave_base) {
  // wave-uniform by construction; readfirstlane makes that visible to the compiler (M0 takes a scalar)
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)(lds_wave_base))));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(gsrc), "s"(lds_addr)
               : "memory");  // M0 is a reserved register: the compiler keeps nothing in it across statements
}

__device__ __forceinline__ void wait_direct_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0) only

struct SyrkParams {
  const uint16_t* xt;  // [Cp / 256][ktiles][256][64]: blocked transposed copy of X (transpose_pad_kernel)
  const uint16_t* x;   // direct form: the activations themselves, [T][C] with T % 64 == 0 and C % 256 == 0
  int64_t Tp, ktiles;
  int32_t C, Cp;
  int32_t tiles_per_side;  // Cp / 256
  int32_t num_tiles;       // upper triangle incl. diagonal
  int32_t splits;          // K splits
  int32_t ktiles_per_split;
  int32_t first_logical;  // work items (tile, split) of earlier launches
  float alpha;
  float* H;             // [C][C]
  float* partial;       // [splits][num_tiles][128*128] when splits > 1
  const int2* tile_map;  // [num_tiles] (bi, bj) in patch-major order
};

// tile_map[l] = l-th tile when the upper triangle is walked patch by patch (8 x 8 tiles; tiles inside a patch
// row-major; diagonal patches keep only bi <= bj).  Whole off-diagonal patches come first, then the ragged
// off-diagonal ones at the matrix edge, then the diagonal patches: a whole patch is 64 tiles, so every aligned
// window of kWindow = 32 consecutive tiles of the first class is a 4 x 8 half patch that needs 12 operand panels.
constexpr int kWindow = 32;  // tiles per window = workgroups of an XCD in a round of one workgroup per compute unit
__global__ void syrk_tile_map_kernel(int2* map, int n, int num_tiles) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= num_tiles) return;
  const int P = (n + kPatch - 1) / kPatch;
  int rem = l;
  for (int cls = 0; cls < 3; ++cls) {
    for (int pi = 0; pi < P; ++pi) {
      const int mi = min(kPatch, n - pi * kPatch);
      for (int pj = pi; pj < P; ++pj) {
        const int mj = min(kPatch, n - pj * kPatch);
        const int kind = pi == pj ? 2 : (mi == kPatch && mj == kPatch ? 0 : 1);
        if (kind != cls) continue;
        const int count = pi == pj ? mi * (mi + 1) / 2 : mi * mj;
        if (rem < count) {
          int i, j;
          if (pi == pj) {
            i = 0;
            while (rem >= mi - i) {
              rem -= mi - i;
              ++i;
            }
            j = i + rem;
          } else {
            i = rem / mj;
            j = rem % mj;
          }
          map[l] = make_int2(pi * kPatch + i, pj * kPatch + j);
          return;
        }
        rem -= count;
      }
    }
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// Transposing LDS read: the 16 lanes of a group read a block of 4 rows x 16 columns of 16-bit elements (lane 4 q + p
// supplies the address of row q, columns 4 p .. 4 p + 3) and lane i receives column i of the 4 rows.
__device__ __forceinline__ uint2 lds_read_transposed(const unsigned char* at) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                       (__attribute__((address_space(3))) s16x4*)(at)));
}

// DIRECT: the operand tiles are staged from the activations as they are, X[token][channel], and the fragments are
// read from LDS with the transposing read, so no transposed copy of X is made.  The LDS image of a k-tile is then
// [64 tokens][256 channels]; the 32-byte segment (16 channels) index of a row is XOR-ed with
// (token % 4) | ((token / 8) % 2) * 4, which puts the 8 rows x 32 bytes that the 32 lanes of a half-wave read at once
// on all 64 banks.
template <int ST, bool DIRECT>
__global__ __launch_bounds__(kSyrkThreads, 1) void syrk_kernel(const SyrkParams p) {
  extern __shared__ __align__(1024) unsigned char lds[];  // [2 buffers][A, B][32 KiB]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;  // wave's 128 x 64 sub-tile
  // XCD-aware renumbering (bijective for any grid size): consecutive logical ids stay on one XCD
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r8 = nwg & 7;
  const int logical =
      p.first_logical + (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (blockIdx.x >> 3);
  // Work items in (split, tile) order, except that the tiles behind the last whole window of kWindow tiles come after
  // the whole windows of all splits: every split then starts on a window boundary, so the workgroups an XCD gets in a
  // round are an aligned window (a 4 x 8 half patch: 12 operand panels) in every split, not only in the first, and
  // the XCDs of a round still work on neighbouring patches of the same range of k (which they share below the L2s).
  int split, tile_idx;
  {
    const int whole = p.num_tiles / kWindow * kWindow, rest = p.num_tiles - whole;
    if (logical < whole * p.splits) {
      split = logical / whole;
      tile_idx = logical - split * whole;
    } else {
      const int l2 = logical - whole * p.splits;
      split = l2 / rest;
      tile_idx = whole + (l2 - split * rest);
    }
  }
  const int2 t = p.tile_map[tile_idx];
  const int bi = t.x, bj = t.y;
  const bool diag = bi == bj;
  // on a diagonal tile the sub-tiles strictly below the diagonal are never stored: their waves only stage
  const bool computes = !diag || wc * 64 + 64 > wr * 128;

  const int64_t k_begin = static_cast<int64_t>(split) * p.ktiles_per_split * kBK;
  int64_t k_end = k_begin + static_cast<int64_t>(p.ktiles_per_split) * kBK;
  if (k_end > p.Tp) k_end = p.Tp;
  const int nk = k_end > k_begin ? static_cast<int>((k_end - k_begin) / kBK) : 0;

  // staging: wave w fills rows [32 w, 32 w + 32) of each operand tile with 4 wave-instructions of 8 rows;
  // lane -> (row = lane / 8, LDS chunk position = lane % 8), global chunk = position ^ ((row >> 1) & 7)
  // (direct form: wave w fills token rows [8 w, 8 w + 8) with 4 wave-instructions of 2 rows x 512 bytes;
  // lane -> (row = lane / 32, 16-byte slot = lane % 32), global segment = (slot / 2) ^ swizzle(row))
  const int st_row = lane >> 3, st_pos = lane & 7;
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
  const int64_t stage_step = DIRECT ? static_cast<int64_t>(kBK) * p.C : static_cast<int64_t>(kTile) * kBK;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if constexpr (DIRECT) {
      const int krow = wave * 8 + it * 2 + (lane >> 5), slot = lane & 31;
      const int chan = (((slot >> 1) ^ ((krow & 3) | (((krow >> 3) & 1) << 2))) << 4) + (slot & 1) * 8;
      a_src[it] = p.x + (k_begin + krow) * p.C + bi * kTile + chan;
      b_src[it] = p.x + (k_begin + krow) * p.C + bj * kTile + chan;
    } else {
      const int row = wave * 32 + it * 8 + st_row;
      const int chunk = st_pos ^ ((row >> 1) & 7);
      a_src[it] = p.xt + ((static_cast<int64_t>(bi) * p.ktiles + k_begin / kBK) * kTile + row) * kBK + chunk * 8;
      b_src[it] = p.xt + ((static_cast<int64_t>(bj) * p.ktiles + k_begin / kBK) * kTile + row) * kBK + chunk * 8;
    }
  }
  // either form: a wave fills 4 KiB of each operand image per k-tile, 1 KiB per wave-instruction
#define DCQ_STAGE(buf, kt)                                                                      \
  do {                                                                                          \
    unsigned char* a_dst = lds + (buf) * 2 * kOperandBytes + wave * 32 * kRowBytes;             \
    unsigned char* b_dst = a_dst + kOperandBytes;                                               \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                          \
      load_to_lds_16(a_src[it] + static_cast<int64_t>(kt) * stage_step, a_dst + it * 8 * kRowBytes);  \
      if (!diag) load_to_lds_16(b_src[it] + static_cast<int64_t>(kt) * stage_step, b_dst + it * 8 * kRowBytes); \
    }                                                                                           \
  } while (0)

  constexpr int MT = 8, NT = 4;  // 16 x 16 tiles per wave: 128 rows x 64 columns
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.0f;

  if (nk > 0) DCQ_STAGE(0, 0);
  wait_direct_loads();
  __syncthreads();
  // fragment addressing (16x16x32): lane -> tile row (lane & 15), k-chunk (4 ks + (lane >> 4)), swizzled position.
  // Every 16-row block starts at a multiple of 16, so the swizzle term (row >> 1) & 7 is the same for all blocks.
  const int frag_row = lane & 15, frag_quarter = lane >> 4;
  const int sw = (frag_row >> 1) & 7;
  const int a_base = (wr * 128 + frag_row) * kRowBytes, b_base = (wc * 64 + frag_row) * kRowBytes;
  // direct form: lane 16 g + 4 q + pp reads token rows 8 g + q (+ 32 per k-half, + 4 for the fragment's second half),
  // 8 bytes at column pp of the block's 32-byte segment; swizzle of those rows: q | (g % 2) * 4
  const int tr_lane = (8 * frag_quarter + (frag_row >> 2)) * 512 + (lane & 3) * 8;
  const int tr_swz = (frag_row >> 2) | ((frag_quarter & 1) << 2);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) DCQ_STAGE(buf ^ 1, kt + 1);
    const unsigned char* a_lds = lds + buf * 2 * kOperandBytes;
    const unsigned char* b_lds = diag ? a_lds : a_lds + kOperandBytes;
    if (computes) {
      // Software pipeline over the 16 (k-half, row-block) groups of this k-tile: the fragment of group g + 1 is
      // requested before the 4 matrix instructions of group g are issued, so its LDS latency passes behind them
      // (the scheduler is pinned to that order: left alone it pairs every read with an immediate wait).
      const int pos0 = ((0 * 4 + frag_quarter) ^ sw) << 4, pos1 = ((1 * 4 + frag_quarter) ^ sw) << 4;
      constexpr int D = 1;  // groups of read-ahead (2 measured no faster)
      constexpr int RD = DIRECT ? 2 : 1;  // LDS reads per fragment
      // fragment of 16-row block `blk` (of the wave's 8 or 4) for k-half ks
      auto read_a = [&](int blk, int ks) __attribute__((always_inline)) -> uint4 {
        if constexpr (DIRECT) {
          const unsigned char* at = a_lds + tr_lane + ks * (32 * 512) + (((wr * 8 + blk) ^ tr_swz) << 5);
          const uint2 lo = lds_read_transposed(at), hi = lds_read_transposed(at + 4 * 512);
          return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
          return *reinterpret_cast<const uint4*>(a_lds + a_base + blk * 16 * kRowBytes + (ks ? pos1 : pos0));
        }
      };
      auto read_b = [&](int blk, int ks) __attribute__((always_inline)) -> uint4 {
        if constexpr (DIRECT) {
          const unsigned char* at = b_lds + tr_lane + ks * (32 * 512) + (((wc * 4 + blk) ^ tr_swz) << 5);
          const uint2 lo = lds_read_transposed(at), hi = lds_read_transposed(at + 4 * 512);
          return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
          return *reinterpret_cast<const uint4*>(b_lds + b_base + blk * 16 * kRowBytes + (ks ? pos1 : pos0));
        }
      };
      uint4 af[D + 1], bf[2][NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) bf[0][n] = read_b(n, 0);
#pragma unroll
      for (int d = 0; d < D; ++d) af[d] = read_a(d, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, RD * (NT + D), 0);
      static_for<2 * MT>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        constexpr int ks = g / MT, m = g % MT;
        if constexpr (g + D < 2 * MT) {
          constexpr int ks1 = (g + D) / MT, m1 = (g + D) % MT;
          af[(g + D) % (D + 1)] = read_a(m1, ks1);
          if constexpr (m1 == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[1][n] = read_b(n, 1);
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = mfma_16x16x32<ST>(af[g % (D + 1)], bf[ks][n], acc[m][n]);
        constexpr int reads = (g + D < 2 * MT) ? ((g + D) % MT == 0 ? 1 + NT : 1) : 0;
        if constexpr (reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, RD * reads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
      });
    }
    wait_direct_loads();  // tile kt + 1 has landed (this wave's pieces) ...
    __syncthreads();      // ... and everyone's; also fences the reads of tile kt
  }
#undef DCQ_STAGE
  if (!computes) return;

  // epilogue.  C/D layout of the 16x16 instruction: col = lane & 15, row = reg + 4 * (lane >> 4)
  const int col_in = lane & 15, row_off = 4 * (lane >> 4);
  if (p.splits > 1) {
    float* slab = p.partial + (static_cast<int64_t>(split) * p.num_tiles + tile_idx) * (kTile * kTile);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr * 128 + m * 16 + r + row_off;
          const int col = wc * 64 + n * 16 + col_in;
          slab[row * kTile + col] = acc[m][n][r];
        }
    return;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bi * kTile + wr * 128 + m * 16 + r + row_off;
        const int col = bj * kTile + wc * 64 + n * 16 + col_in;
        if (row < p.C && col < p.C) {
          const float v = p.alpha * acc[m][n][r];
          if (!diag || col >= row) {
            float* h = p.H + static_cast<int64_t>(row) * p.C + col;
            const float sum = *h + v;
            *h = sum;
            if (col != row) p.H[static_cast<int64_t>(col) * p.C + row] = sum;
          }
        }
      }
}

// H[tile] += alpha * sum_s partial[s][tile]  (fixed order), mirrored.  16 workgroups per tile, 16 rows each;
// one thread sums 4 consecutive columns (16-byte slab reads).
constexpr int kReduceRows = 16;

__global__ __launch_bounds__(256) void syrk_reduce_kernel(const SyrkParams p) {
  const int tile_idx = blockIdx.x / (kTile / kReduceRows);
  const int r0 = (blockIdx.x % (kTile / kReduceRows)) * kReduceRows;
  const int2 t = p.tile_map[tile_idx];
  const int bi = t.x, bj = t.y;
  // Off-diagonal tiles that lie wholly inside the matrix (all but the diagonal and the ragged edge): 16-byte reads and
  // writes of H, and the mirrored strip (256 rows x 16 columns) goes through LDS so that every thread writes one
  // row's 64 contiguous bytes instead of sixteen 4-byte words a whole row apart.  The same sums, bit for bit.
  __shared__ __align__(16) float strip[kReduceRows][kTile];
  if (bi != bj && (bi + 1) * kTile <= p.C && (bj + 1) * kTile <= p.C && (p.C & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(p.H) & 15u) == 0) {
    for (int e = threadIdx.x; e < kReduceRows * (kTile / 4); e += 256) {
      const int r = e / (kTile / 4), c = (e % (kTile / 4)) * 4;
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sp = 0; sp < p.splits; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(
            p.partial + (static_cast<int64_t>(sp) * p.num_tiles + tile_idx) * (kTile * kTile) + (r0 + r) * kTile + c);
        s4.x = s4.x + v.x;
        s4.y = s4.y + v.y;
        s4.z = s4.z + v.z;
        s4.w = s4.w + v.w;
      }
      float4* h = reinterpret_cast<float4*>(p.H + static_cast<int64_t>(bi * kTile + r0 + r) * p.C + bj * kTile + c);
      float4 sum = *h;
      sum.x = sum.x + p.alpha * s4.x;
      sum.y = sum.y + p.alpha * s4.y;
      sum.z = sum.z + p.alpha * s4.z;
      sum.w = sum.w + p.alpha * s4.w;
      *h = sum;
      *reinterpret_cast<float4*>(&strip[r][c]) = sum;
    }
    __syncthreads();
    float* mirror = p.H + static_cast<int64_t>(bj * kTile + threadIdx.x) * p.C + bi * kTile + r0;
#pragma unroll
    for (int k = 0; k < kReduceRows; k += 4)
      *reinterpret_cast<float4*>(mirror + k) = make_float4(strip[k][threadIdx.x], strip[k + 1][threadIdx.x],
                                                           strip[k + 2][threadIdx.x], strip[k + 3][threadIdx.x]);
    return;
  }
  for (int e = threadIdx.x; e < kReduceRows * (kTile / 4); e += 256) {
    const int r = r0 + e / (kTile / 4), c = (e % (kTile / 4)) * 4;
    const int row = bi * kTile + r, col = bj * kTile + c;
    if (row >= p.C || col >= p.C) continue;
    if (bi == bj && c + 3 < r) continue;  // entirely below the diagonal
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < p.splits; ++sp) {
      const float4 v = *reinterpret_cast<const float4*>(
          p.partial + (static_cast<int64_t>(sp) * p.num_tiles + tile_idx) * (kTile * kTile) + r * kTile + c);
      s4.x = s4.x + v.x;
      s4.y = s4.y + v.y;
      s4.z = s4.z + v.z;
      s4.w = s4.w + v.w;
    }
    const float s[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cc = col + i;
      if (cc >= p.C || (bi == bj && cc < row)) continue;
      float* h = p.H + static_cast<int64_t>(row) * p.C + cc;
      const float sum = *h + p.alpha * s[i];
      *h = sum;
      if (cc != row) p.H[static_cast<int64_t>(cc) * p.C + row] = sum;
    }
  }
}


static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct HessianPlan {
  int64_t Cp, Tp, ktiles;
  int tiles_side, num_tiles, splits, ktiles_per_split;
  size_t xt_bytes, map_bytes, partial_bytes;
};

static HessianPlan plan_hessian(int64_t T, int64_t C) {
  HessianPlan h;
  h.Cp = round_up(C, kTile);
  h.Tp = round_up(T, kBK);
  h.ktiles = h.Tp / kBK;  // k-tiles per panel (blocked layout, see transpose_pad_kernel)
  h.tiles_side = static_cast<int>(h.Cp / kTile);
  h.num_tiles = h.tiles_side * (h.tiles_side + 1) / 2;
  const int ktiles = static_cast<int>(h.ktiles);
  // one workgroup per compute unit: pick the K split whose workgroup count fills whole rounds of 256 best,
  // with at least 32 k-tiles per split and at most 1.5 GB of partial slabs
  int splits = 1;
  double best = 0.0;
  for (int sp = 1; sp <= 64; ++sp) {
    if (sp > 1 && ktiles / sp < 32) break;
    const int64_t wgs = static_cast<int64_t>(h.num_tiles) * sp;
    const int64_t rounds = (wgs + 255) / 256;
    if (sp > 1 && wgs * kTile * kTile * 4 > (3ll << 29)) break;
    const double eff = static_cast<double>(wgs) / static_cast<double>(rounds * 256);
    if (eff > best + 0.02) {
      best = eff;
      splits = sp;
    }
  }
  h.splits = splits;
  h.ktiles_per_split = (ktiles + splits - 1) / splits;
  h.xt_bytes = static_cast<size_t>(round_up(h.Cp * h.Tp * 2, 256));
  h.map_bytes = static_cast<size_t>(round_up(static_cast<int64_t>(h.num_tiles) * 8, 256));
  h.partial_bytes = splits > 1 ? static_cast<size_t>(splits) * h.num_tiles * kTile * kTile * 4 : 0;
  return h;
}

// =============================================================================================
// GPTQ column sweep of one block
// =============================================================================================
constexpr int kSweepCols = 128;     // columns per block (the reference's block_size)
constexpr int kSweepTeam = 16;      // lanes per row
constexpr int kSweepPerLane = 8;    // columns per lane
constexpr int kSweepThreads = 512;  // largest workgroup: 8 waves = 32 rows; the launch picks 64..512 from the row count

struct SweepParams {
  dcq_step step;
  float* W;        // [R][ldw], columns already permuted; only read here
  int64_t ldw;
  const float* U;  // [C][C] upper triangular, permuted order
  int64_t C;
  int64_t R;
  int32_t c0, bc;       // block start / width (<= 128)
  const float* scale;   // [row_groups][col_groups]
  const float* zero;    // same shape, or NULL when zero_scalar applies
  float zero_scalar;
  int32_t row_group_size, col_group_size, col_groups;
  const int64_t* perm;  // [C] original column of each permuted column
  int32_t has_bound_min, has_bound_max;
  float bound_min, bound_max;
  float* Q;    // [R][ldw] codes, permuted order
  float* Err;  // [R][ld_err], 128 columns used
  int64_t ld_err;
  int32_t u_quads;  // the block of U may be read as aligned float4
  int32_t fast_layout;  // Q and Err rows of the block take aligned float4 stores
  int32_t rows_per_block;  // rows a workgroup sweeps at a time (its first 16 * rows_per_block threads; the rest only stage)
};

#ifdef DCQ_STAMP
__device__ unsigned long long g_stamp[4];  // timing-only builds: shader cycles / 100 MHz ticks of one wave's column loop
#endif

// DPP moves inside every 16-lane row: the value of the row's lane 0 to all its lanes (row_newbcast:0), and a rotation
// by one lane towards lane 0 (lane l takes lane l + 1's value, lane 15 takes lane 0's: row_ror:15).  No LDS round trip.
__device__ __forceinline__ float row_first(float v) {
  const int bits = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(bits, bits, 0x150, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_rotate_down(float v) {
  const int bits = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(bits, bits, 0x12F, 0xf, 0xf, false));
}

// The guard maximum of a column as volatile assembly, so it stays inside its column: left to the compiler such
// by-products are sunk to the end of the chain, which keeps their inputs alive and spills them to scratch memory.
__device__ __forceinline__ void track_max_abs(float& worst, float a, float b) {
  asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(worst) : "v"(a), "v"(b));
}

// One row per 16-lane team, the block's 128 columns in registers (8 per lane).  Per column the serial chain is
//   clamp -> (+ zero) -> / scale -> round -> clamp -> dequantize -> error / U[c][c] -> broadcast -> update,
// and a wave has nothing else to do while it runs, so its length in dependent instructions is the kernel's time.
//
// FAST (whole 128-column blocks, aligned buffers): both divisions use the hoisted-reciprocal form (bit-identical to
// IEEE division inside its guard, see make_divisor) and the broadcast is a DPP move.  A DPP move names its source
// lane in the instruction, so instead of unrolling all 128 columns (the straight-line code then outgrows the
// instruction cache and the wave waits for its own instructions) the lanes' columns rotate: after a lane's 8 columns
// are done every lane hands its state to its lower neighbour, and the lane whose columns are swept is always lane 0.
// Lane 0 stores the 8 codes and errors of each step straight to memory.  The guard on the dividends (|x| <= 2^60) is
// checked after the fact: the caller reruns the rows through the plain form when `overflow` comes back set, which
// real weights never do.
// The lane's 8 weights of row rr in this block with their scales / zeros (looked up through the act-order permutation).
struct RowState {
  float w[kSweepPerLane], s[kSweepPerLane], z[kSweepPerLane];
};

template <int ZD>
__device__ __forceinline__ RowState load_row_state(const SweepParams& p, int64_t rr, int tl) {
  RowState r;
  const int64_t rg = rr / p.row_group_size;
#pragma unroll
  for (int i = 0; i < kSweepPerLane; ++i) {
    const int col = tl * kSweepPerLane + i;
    const bool in = col < p.bc;
    r.w[i] = in ? p.W[rr * p.ldw + p.c0 + col] : 0.0f;
    const int64_t cg = in ? p.perm[p.c0 + col] / p.col_group_size : 0;
    r.s[i] = p.scale[rg * p.col_groups + cg];
    r.z[i] = ZD == DCQ_ZERO_NONE ? 0.0f : (p.zero ? p.zero[rg * p.col_groups + cg] : p.zero_scalar);
  }
  return r;
}

template <int ZD, bool INT_Q, bool FAST>
__device__ __forceinline__ void sweep_rows(const SweepParams& p, const float* u_lds, const float* diag_s,
                                           const float* diag_y, const float* cb, const RowState& state, int64_t rr,
                                           bool live, int tl, float (&q)[kSweepPerLane], float (&er)[kSweepPerLane],
                                           bool& overflow) {
  const dcq_step& st = p.step;
  float w[kSweepPerLane], s[kSweepPerLane], z[kSweepPerLane];
#pragma unroll
  for (int i = 0; i < kSweepPerLane; ++i) {
    w[i] = state.w[i];
    s[i] = state.s[i];
    z[i] = state.z[i];
    q[i] = 0.0f;
    er[i] = 0.0f;
  }
  if (FAST) {
    float y[kSweepPerLane];  // refined reciprocals of the scales
    float worst = 0.0f;      // largest dividend seen by this lane
    bool scales_ok = true;
#pragma unroll
    for (int i = 0; i < kSweepPerLane; ++i) {
      const Divisor d = make_divisor(s[i], 0.0f);
      y[i] = d.y;
      scales_ok = scales_ok && d.fast;
    }
    // RangeBound as one median: an absent bound is an infinity, which leaves every finite value alone
    const float bound_lo = p.has_bound_min ? p.bound_min : -INFINITY, bound_hi = p.has_bound_max ? p.bound_max : INFINITY;
    float* q_out = p.Q + rr * p.ldw + p.c0;
    float* e_out = p.Err + rr * p.ld_err;
#ifdef DCQ_STAMP
    const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int step = 0; step < kSweepTeam; ++step) {
      // this lane now holds the columns of lane (tl + step) mod 16; row c of U, those 8 entries
      const float* u_lane = u_lds + (step * kSweepPerLane) * kSweepCols + ((tl + step) & (kSweepTeam - 1)) * kSweepPerLane;
      // all LDS operands of the step's 8 columns in one burst (one wait per step instead of one inside every column's
      // chain): the 8 rows of U and the reciprocals of their diagonal entries
      float4 ur[kSweepPerLane][2];
#pragma unroll
      for (int i = 0; i < kSweepPerLane; ++i) {
        ur[i][0] = *reinterpret_cast<const float4*>(u_lane + i * kSweepCols);
        ur[i][1] = *reinterpret_cast<const float4*>(u_lane + i * kSweepCols + 4);
      }
      const float4 yd0 = *reinterpret_cast<const float4*>(diag_y + step * kSweepPerLane);
      const float4 yd1 = *reinterpret_cast<const float4*>(diag_y + step * kSweepPerLane + 4);
      const float yd[8] = {yd0.x, yd0.y, yd0.z, yd0.w, yd1.x, yd1.y, yd1.z, yd1.w};
      float qv[kSweepPerLane], ev[kSweepPerLane];
#pragma unroll
      for (int i = 0; i < kSweepPerLane; ++i) {
        float a = __builtin_amdgcn_fmed3f(w[i], bound_lo, bound_hi);
        if (ZD == DCQ_ZERO_POST_SCALE) a = a + z[i];
        Divisor dq;
        dq.s = s[i];
        dq.y = y[i];
        float t = div_fast(a, dq);
        if (ZD == DCQ_ZERO_PRE_SCALE) t = t + z[i];
        const float qi = INT_Q ? __builtin_amdgcn_fmed3f(rintf(t), st.tensor.lo, st.tensor.hi)
                               : apply_rounding(st.tensor, t, cb);
        const float d = from_code<ZD>(qi, s[i], z[i]);
        const float diff = w[i] - d;
        track_max_abs(worst, a, diff);
        // in lane 0 entry i of its 8 entries of row c is the diagonal U[c][c]: the error that lane forms is the one
        // to broadcast (the other lanes' quotients are never looked at)
        const float4 u0 = ur[i][0], u1 = ur[i][1];
        const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        Divisor dc;
        dc.s = uu[i];
        dc.y = yd[i];
        const float e_own = div_fast(diff, dc);
        qv[i] = qi;
        ev[i] = e_own;
        const float e = row_first(e_own);
        // every lane updates: columns at or left of c are finished (their codes and errors are taken), so what
        // happens to them no longer matters.  One fused multiply-add per element (two elements per instruction)
        const f32x2 ne = {-e, -e};
#pragma unroll
        for (int k = 0; k < kSweepPerLane; k += 2) {
          const f32x2 wk = {w[k], w[k + 1]}, uk = {uu[k], uu[k + 1]};
          const f32x2 r = __builtin_elementwise_fma(ne, uk, wk);
          w[k] = r.x;
          w[k + 1] = r.y;
        }
      }
      if (tl == 0 && live) {
        *reinterpret_cast<float4*>(q_out + step * kSweepPerLane) = make_float4(qv[0], qv[1], qv[2], qv[3]);
        *reinterpret_cast<float4*>(q_out + step * kSweepPerLane + 4) = make_float4(qv[4], qv[5], qv[6], qv[7]);
        *reinterpret_cast<float4*>(e_out + step * kSweepPerLane) = make_float4(ev[0], ev[1], ev[2], ev[3]);
        *reinterpret_cast<float4*>(e_out + step * kSweepPerLane + 4) = make_float4(ev[4], ev[5], ev[6], ev[7]);
      }
#pragma unroll
      for (int k = 0; k < kSweepPerLane; ++k) {
        w[k] = row_rotate_down(w[k]);
        s[k] = row_rotate_down(s[k]);
        y[k] = row_rotate_down(y[k]);
        if (ZD != DCQ_ZERO_NONE) z[k] = row_rotate_down(z[k]);
      }
    }
#ifdef DCQ_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_stamp[0] = __builtin_amdgcn_s_memtime() - st_c0;
      g_stamp[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
    overflow = !scales_ok || !(worst <= 0x1p60f);
    return;
  }
  for (int owner = 0; owner < kSweepTeam; ++owner) {
    if (owner * kSweepPerLane >= p.bc) break;  // uniform
#pragma unroll
    for (int i = 0; i < kSweepPerLane; ++i) {
      const int c = owner * kSweepPerLane + i;
      if (c < p.bc) {  // uniform
        // every lane evaluates its own slot i; only the owner's value is used
        float x = w[i];
        if (p.has_bound_min) x = fmaxf(x, p.bound_min);
        if (p.has_bound_max) x = fminf(x, p.bound_max);
        const float t = to_code_input<ZD>(x, s[i], z[i]);
        const float qi = INT_Q ? round_int(t, st.tensor.lo, st.tensor.hi) : apply_rounding(st.tensor, t, cb);
        const float d = from_code<ZD>(qi, s[i], z[i]);
        const float ucc = diag_s[c];
        const float e_own = (w[i] - d) / ucc;
        if (tl == owner) {
          q[i] = qi;
          er[i] = e_own;
        }
        const float e = __shfl(e_own, owner, kSweepTeam);
        if (tl >= owner) {
          const float4 u0 = *reinterpret_cast<const float4*>(u_lds + c * kSweepCols + tl * kSweepPerLane);
          const float4 u1 = *reinterpret_cast<const float4*>(u_lds + c * kSweepCols + tl * kSweepPerLane + 4);
          const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
          for (int k = 0; k < kSweepPerLane; ++k) {
            // columns <= c of the owner lane see u = 0 below the diagonal except the diagonal itself,
            // whose update is harmless (that column is finished); fused, as in the fast form
            w[k] = __builtin_fmaf(-e, uu[k], w[k]);
          }
        }
      }
    }
  }
  overflow = false;
}

template <int ZD, bool INT_Q>
__global__ __launch_bounds__(kSweepThreads, 4) void gptq_sweep_kernel(const SweepParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef DCQ_STAMP
  const uint64_t st_k0 = __builtin_amdgcn_s_memrealtime();
#endif
  float* u_lds = reinterpret_cast<float*>(smem);           // [128][128], rows >= bc / cols >= bc zero
  float* cb = u_lds + kSweepCols * kSweepCols;             // tensor codebook [256]
  float* diag_s = cb + 256;                                // U[c][c] of the block's columns [128]
  float* diag_y = diag_s + kSweepCols;                     // their refined reciprocals [128]
  int* diag_ok = reinterpret_cast<int*>(diag_y + kSweepCols);  // [1]: every diagonal entry is inside the fast guard
  // stage the diagonal block of U four columns at a time; entries left of the diagonal are never used with a
  // non-zero meaning, so quads wholly below it are written as zeros without touching memory
  const int threads = static_cast<int>(blockDim.x);
  const bool quads = p.u_quads != 0;
  const int tl = threadIdx.x & (kSweepTeam - 1);
  const int team = threadIdx.x / kSweepTeam;
  const int teams_per_block = p.rows_per_block;
  // the first rows' weights, and their scales behind two dependent look-ups, are requested before the block of U is
  // staged, so their memory latency passes behind the staging
  const bool sweeps = team < teams_per_block;  // (whole waves: rows_per_block is a multiple of 4)
  const int64_t first_row = static_cast<int64_t>(blockIdx.x) * teams_per_block + team;
  RowState state = load_row_state<ZD>(p, sweeps && first_row < p.R ? first_row : 0, tl);
  if (threadIdx.x == 0) *diag_ok = 1;
#pragma unroll 8
  for (int e = threadIdx.x; e < kSweepCols * kSweepCols / 4; e += threads) {
    const int r = e / (kSweepCols / 4), c = (e % (kSweepCols / 4)) * 4;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r < p.bc && c + 3 >= r && c < p.bc) {
      const float* src = p.U + (p.c0 + r) * p.C + p.c0 + c;
      if (quads && c + 3 < p.bc) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = src[0];
        if (c + 1 < p.bc) v.y = src[1];
        if (c + 2 < p.bc) v.z = src[2];
        if (c + 3 < p.bc) v.w = src[3];
      }
      if (c < r) v.x = 0.0f;
      if (c + 1 < r) v.y = 0.0f;
      if (c + 2 < r) v.z = 0.0f;
    }
    *reinterpret_cast<float4*>(u_lds + r * kSweepCols + c) = v;
  }
  if (!INT_Q)
    for (int i = threadIdx.x; i < p.step.tensor.codebook_size; i += threads) cb[i] = p.step.tensor.codebook[i];
  __syncthreads();
  for (int c = threadIdx.x; c < kSweepCols; c += threads) {
    const float ucc = c < p.bc ? u_lds[c * kSweepCols + c] : 1.0f;
    const Divisor d = make_divisor(ucc, 0.0f);
    diag_s[c] = ucc;
    diag_y[c] = d.y;
    if (!d.fast) *diag_ok = 0;  // (same value from every writer)
  }
  __syncthreads();
  const bool fast_block = *diag_ok != 0 && p.bc == kSweepCols && p.fast_layout != 0;
#ifdef DCQ_STAMP
  if (blockIdx.x == 0 && threadIdx.x == 0) g_stamp[2] = __builtin_amdgcn_s_memrealtime() - st_k0;
#endif

  if (!sweeps) return;  // staging helpers are done (no barrier follows)
  for (int64_t row0 = static_cast<int64_t>(blockIdx.x) * teams_per_block; row0 < p.R;
       row0 += static_cast<int64_t>(gridDim.x) * teams_per_block) {
    const int64_t row = row0 + team;
    const bool live = row < p.R;
    const int64_t rr = live ? row : 0;
    if (row != first_row) state = load_row_state<ZD>(p, rr, tl);
    float q[kSweepPerLane], er[kSweepPerLane];
    bool overflow = true;
    if (fast_block) sweep_rows<ZD, INT_Q, true>(p, u_lds, diag_s, diag_y, cb, state, rr, live, tl, q, er, overflow);
#ifdef DCQ_STAMP
    if (blockIdx.x == 0 && threadIdx.x == 0) g_stamp[3] = __builtin_amdgcn_s_memrealtime() - st_k0;
#endif
    if (!__any(overflow)) continue;  // (uniform over the wave) the fast form has stored its results itself
    sweep_rows<ZD, INT_Q, false>(p, u_lds, diag_s, diag_y, cb, state, rr, live, tl, q, er, overflow);
    if (live) {
#pragma unroll
      for (int i = 0; i < kSweepPerLane; ++i) {
        const int col = tl * kSweepPerLane + i;
        if (col < p.bc) p.Q[row * p.ldw + p.c0 + col] = q[i];
        p.Err[row * p.ld_err + col] = col < p.bc ? er[i] : 0.0f;
      }
    }
  }
}

// =============================================================================================
// Column gather of a row-major float32 matrix: dst[r][c] = keep(index[c]) ? src[r][index[c]] : 0
// =============================================================================================
// One workgroup per row at a time: the row is staged in LDS with coalesced 16-byte loads (columns flagged in
// `zero_columns` are zeroed on the way in), gathered from there, and written with coalesced 16-byte stores; a
// scattered 4-byte read from global memory would move a whole cache line per element.
constexpr int kGatherThreads = 256;

template <int ST>
__global__ __launch_bounds__(kGatherThreads) void gather_columns_kernel(const void* src_any, int64_t ld_src, int64_t R,
                                                                         int32_t C, const int64_t* index,
                                                                         const uint8_t* zero_columns, float* dst,
                                                                         int64_t ld_dst) {
  using St = Storage<ST>;
  using elem_t = typename St::elem_t;
  extern __shared__ __align__(16) unsigned char gather_smem[];
  float* row = reinterpret_cast<float*>(gather_smem);
  const elem_t* src = static_cast<const elem_t*>(src_any);
  const int c4 = C & ~3;
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    const elem_t* s = src + r * ld_src;
    for (int c = threadIdx.x * 4; c < c4; c += kGatherThreads * 4) {
      float4 v;
      if constexpr (ST == DCQ_F32) {
        v = *reinterpret_cast<const float4*>(s + c);
      } else {  // four 16-bit values, widened on the way in
        const uint2 raw = *reinterpret_cast<const uint2*>(s + c);
        v = make_float4(St::load(static_cast<uint16_t>(raw.x & 0xffffu)), St::load(static_cast<uint16_t>(raw.x >> 16)),
                        St::load(static_cast<uint16_t>(raw.y & 0xffffu)), St::load(static_cast<uint16_t>(raw.y >> 16)));
      }
      if (zero_columns) {
        const uchar4 z = *reinterpret_cast<const uchar4*>(zero_columns + c);
        if (z.x) v.x = 0.0f;
        if (z.y) v.y = 0.0f;
        if (z.z) v.z = 0.0f;
        if (z.w) v.w = 0.0f;
      }
      *reinterpret_cast<float4*>(row + c) = v;
    }
    for (int c = c4 + threadIdx.x; c < C; c += kGatherThreads)
      row[c] = zero_columns && zero_columns[c] ? 0.0f : St::load(s[c]);
    __syncthreads();
    float* d = dst + r * ld_dst;
    for (int c = threadIdx.x * 4; c < c4; c += kGatherThreads * 4) {
      const longlong2 i01 = *reinterpret_cast<const longlong2*>(index + c);
      const longlong2 i23 = *reinterpret_cast<const longlong2*>(index + c + 2);
      *reinterpret_cast<float4*>(d + c) = make_float4(row[i01.x], row[i01.y], row[i23.x], row[i23.y]);
    }
    for (int c = c4 + threadIdx.x; c < C; c += kGatherThreads) d[c] = row[index[c]];
    __syncthreads();
  }
}

}  // namespace dcq

using namespace dcq;

extern "C" {

#ifdef DCQ_STAMP
int dcq_debug_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcq::g_stamp), 32) == hipSuccess ? 0 : -3;
}
#endif

int64_t dcq_hessian_workspace_bytes(int64_t T, int64_t C) {
  if (T <= 0 || C <= 0) return 0;
  const HessianPlan h = plan_hessian(T, C);
  return static_cast<int64_t>(h.xt_bytes + h.map_bytes + h.partial_bytes);
}

int dcq_hessian_accumulate(const void* x, int32_t storage, int64_t T, int64_t C, float alpha, float* H, void* workspace,
                           int64_t workspace_bytes, dcq_stream_t stream) {
  return dcq_hessian_accumulate_ex(x, storage, T, C, alpha, H, workspace, workspace_bytes, 0, nullptr, nullptr, stream);
}

int dcq_hessian_accumulate_ex(const void* x, int32_t storage, int64_t T, int64_t C, float alpha, float* H,
                              void* workspace, int64_t workspace_bytes, int32_t round_width, void* start_event,
                              void* stop_event, dcq_stream_t stream) {
  hipEvent_t prof_start = static_cast<hipEvent_t>(start_event), prof_stop = static_cast<hipEvent_t>(stop_event);
  DCQ_REQUIRE(T >= 0 && C > 0, DCQ_ERR_ARGUMENT, "bad sizes T=%lld C=%lld", (long long)T, (long long)C);
  if (T == 0) return DCQ_OK;
  DCQ_REQUIRE(x && H && workspace, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(storage == DCQ_F16 || storage == DCQ_BF16, DCQ_ERR_UNSUPPORTED,
              "Hessian accumulation takes float16 / bfloat16 activations");
  DCQ_REQUIRE(C < (1 << 20), DCQ_ERR_UNSUPPORTED, "C too large");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(x) & 1u) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
              DCQ_ERR_ARGUMENT, "workspace must be 256-byte aligned");
  const HessianPlan h = plan_hessian(T, C);
  DCQ_REQUIRE(workspace_bytes >= dcq_hessian_workspace_bytes(T, C), DCQ_ERR_ARGUMENT,
              "workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
              (long long)dcq_hessian_workspace_bytes(T, C));
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint16_t* xt = static_cast<uint16_t*>(workspace);
  int2* tile_map = reinterpret_cast<int2*>(static_cast<unsigned char*>(workspace) + h.xt_bytes);
  float* partial = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + h.xt_bytes + h.map_bytes);
  // whole 256-channel panels and whole 64-token k-tiles: the SYRK reads the activations as they are.
  // Up to kDirectMaxC channels, that is; beyond, the direct form's 512-byte pieces (one per token row, rows tens of
  // kilobytes apart) cost the SYRK more than the transposed copy's 32 KiB blocks save (C = 14336, T = 262144: the SYRK
  // alone 45 -> 50 ms against 3 ms of transposing; C = 4096: 4.0 -> 4.1 ms against 0.9 ms).
  const bool direct =
      C % kTile == 0 && T % kBK == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && C <= kDirectMaxC;
  if (!direct) {
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    DCQ_REQUIRE(aligned || (C % 8) != 0, DCQ_ERR_ARGUMENT, "x must be 16-byte aligned");
    dim3 grid(static_cast<unsigned>(h.Tp / kTrTile), static_cast<unsigned>(h.Cp / kTrTile));
    hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, s, static_cast<const uint16_t*>(x), T, C, xt, h.ktiles);
    DCQ_LAUNCH_CHECK("transpose_pad_kernel launch");
  }
  hipLaunchKernelGGL(syrk_tile_map_kernel, dim3((h.num_tiles + 255) / 256), dim3(256), 0, s, tile_map, h.tiles_side,
                     h.num_tiles);
  DCQ_LAUNCH_CHECK("syrk_tile_map_kernel launch");
  SyrkParams p{};
  p.tile_map = tile_map;
  p.xt = xt;
  p.x = static_cast<const uint16_t*>(x);
  p.Tp = h.Tp;
  p.ktiles = h.ktiles;
  p.C = static_cast<int32_t>(C);
  p.Cp = static_cast<int32_t>(h.Cp);
  p.tiles_per_side = h.tiles_side;
  p.num_tiles = h.num_tiles;
  p.splits = h.splits;
  p.ktiles_per_split = h.ktiles_per_split;
  p.alpha = alpha;
  p.H = H;
  p.partial = partial;
  const size_t lds = 2 * 2 * kOperandBytes;
  auto kern = storage == DCQ_BF16 ? (direct ? syrk_kernel<DCQ_BF16, true> : syrk_kernel<DCQ_BF16, false>)
                                  : (direct ? syrk_kernel<DCQ_F16, true> : syrk_kernel<DCQ_F16, false>);
  if (lds > 64 * 1024) {
    int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)),
                       "hipFuncSetAttribute(syrk LDS)");
    if (rc != DCQ_OK) return rc;
  }
  if (prof_start) (void)hipEventRecord(prof_start, s);
  // One launch per round of (tile, split) work items, as many as the chip has compute units (one workgroup each).
  // All workgroups of a launch start together and do the same amount of work, so the 32 of an XCD walk their
  // operand panels in step and each 128-byte line is fetched into that XCD's L2 once for all of them.  In one big
  // launch the workgroups of later rounds start whenever a compute unit frees up, drift apart along k, and re-fetch
  // each other's lines from memory (measured on C = 14336, T = 262144: 1120 -> 1217 TFLOP/s executed).
  const int total = h.num_tiles * h.splits;
  int dev = 0, per_launch = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8)
      per_launch = cus / 8 * 8;  // whole groups of 8: workgroup i runs on XCD i mod 8
  }
  if (round_width >= 8 && round_width < per_launch) per_launch = round_width / 8 * 8;
  for (int first = 0; first < total; first += per_launch) {
    p.first_logical = first;
    const int n = total - first < per_launch ? total - first : per_launch;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(n)), dim3(kSyrkThreads), lds, s, p);
  }
  if (prof_stop) (void)hipEventRecord(prof_stop, s);
  DCQ_LAUNCH_CHECK("syrk_kernel launch");
  if (h.splits > 1) {
    hipLaunchKernelGGL(syrk_reduce_kernel, dim3(static_cast<unsigned>(h.num_tiles) * (kTile / kReduceRows)), dim3(256), 0, s,
                       p);
    DCQ_LAUNCH_CHECK("syrk_reduce_kernel launch");
  }
  return DCQ_OK;
}

int dcq_gather_columns(const void* src, int32_t storage, int64_t ld_src, int64_t R, int64_t C, const int64_t* index,
                       const uint8_t* zero_columns, float* dst, int64_t ld_dst, dcq_stream_t stream) {
  DCQ_REQUIRE(R >= 0 && C > 0 && ld_src >= C && ld_dst >= C, DCQ_ERR_ARGUMENT, "bad sizes");
  DCQ_REQUIRE(storage == DCQ_F32 || storage == DCQ_F16 || storage == DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage type");
  if (R == 0) return DCQ_OK;
  DCQ_REQUIRE(src && index && dst, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(src != static_cast<const void*>(dst), DCQ_ERR_ARGUMENT, "the gather is not done in place");
  DCQ_REQUIRE(C <= 36864, DCQ_ERR_UNSUPPORTED, "rows of more than 36864 columns do not fit the LDS staging");
  DCQ_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                reinterpret_cast<uintptr_t>(index)) & 15u) == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 &&
                  (zero_columns == nullptr || (reinterpret_cast<uintptr_t>(zero_columns) & 3u) == 0),
              DCQ_ERR_ARGUMENT, "src, dst, index must be 16-byte aligned with leading dimensions that are multiples of 4");
  const size_t lds = static_cast<size_t>((C + 3) / 4 * 4) * sizeof(float);
  auto kern = storage == DCQ_F32 ? gather_columns_kernel<DCQ_F32>
                                 : (storage == DCQ_F16 ? gather_columns_kernel<DCQ_F16> : gather_columns_kernel<DCQ_BF16>);
  if (lds > 64 * 1024) {
    int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)),
                       "hipFuncSetAttribute(gather LDS)");
    if (rc != DCQ_OK) return rc;
  }
  const int64_t per_cu = lds > 0 ? (160 * 1024) / static_cast<int64_t>(lds) : 1;
  int64_t blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
  if (blocks > R) blocks = R;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kGatherThreads), lds,
                     static_cast<hipStream_t>(stream), src, ld_src, R, static_cast<int32_t>(C), index, zero_columns, dst,
                     ld_dst);
  DCQ_LAUNCH_CHECK("gather_columns_kernel launch");
  return DCQ_OK;
}

int dcq_gptq_sweep_block(const dcq_step* step, float* W, int64_t ldw, int64_t R, int64_t C, const float* U, int32_t c0,
                         int32_t bc, const float* scale, const float* zero, float zero_scalar, int32_t row_group_size,
                         int32_t col_group_size, int32_t col_groups, const int64_t* perm, float* Q, float* Err,
                         int64_t ld_err, dcq_stream_t stream) {
  DCQ_REQUIRE(step != nullptr, DCQ_ERR_ARGUMENT, "step is NULL");
  DCQ_REQUIRE(step->tensor.kind == DCQ_ROUND_INT || step->tensor.kind == DCQ_ROUND_CODEBOOK, DCQ_ERR_UNSUPPORTED,
              "tensor dtype must be an integer or a codebook type");
  DCQ_REQUIRE(R >= 0 && C > 0 && ldw >= C, DCQ_ERR_ARGUMENT, "bad sizes");
  if (R == 0) return DCQ_OK;
  DCQ_REQUIRE(W && U && scale && perm && Q && Err, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE(bc >= 1 && bc <= kSweepCols && c0 >= 0 && c0 + bc <= C, DCQ_ERR_ARGUMENT,
              "bad block [%d, %d) for C=%lld (at most 128 columns per call)", c0, c0 + bc, (long long)C);
  DCQ_REQUIRE(row_group_size >= 1 && col_group_size >= 1 && col_groups >= 1, DCQ_ERR_ARGUMENT, "bad group geometry");
  DCQ_REQUIRE(ld_err >= kSweepCols, DCQ_ERR_ARGUMENT, "ld_err must be at least 128");
  SweepParams p{};
  p.step = *step;
  p.W = W;
  p.ldw = ldw;
  p.U = U;
  p.C = C;
  p.R = R;
  p.c0 = c0;
  p.bc = bc;
  p.scale = scale;
  p.zero = zero;
  p.zero_scalar = zero_scalar;
  p.row_group_size = row_group_size;
  p.col_group_size = col_group_size;
  p.col_groups = col_groups;
  p.perm = perm;
  p.has_bound_min = step->has_bound_min;
  p.has_bound_max = step->has_bound_max;
  p.bound_min = step->bound_min;
  p.bound_max = step->bound_max;
  p.Q = Q;
  p.Err = Err;
  p.ld_err = ld_err;
  const size_t lds = kSweepCols * kSweepCols * 4 + 256 * 4 + 2 * kSweepCols * 4 + 16;
  p.u_quads = (reinterpret_cast<uintptr_t>(U) & 15u) == 0 && C % 4 == 0 && c0 % 4 == 0;
  p.fast_layout = ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(Err)) & 15u) == 0 && ldw % 4 == 0 &&
                  ld_err % 4 == 0 && c0 % 4 == 0;
  // One column step costs a wave the same whether 4 or 1 of its rows are live, and the 64 KiB block of U lets two
  // workgroups share a CU: spread the rows over 512 workgroups (two per CU) of whole waves, at most 8 waves each.
  // Up to 8192 rows one workgroup per CU does as well (a wave per SIMD either way) and stages the block with twice
  // the threads.
  int64_t rows_per_block = R <= 8192 ? (R + 255) / 256 : (R + 511) / 512;
  rows_per_block = (rows_per_block + 3) / 4 * 4;
  if (rows_per_block > kSweepThreads / kSweepTeam) rows_per_block = kSweepThreads / kSweepTeam;
  // at least 4 waves per workgroup: with fewer rows the extra waves only help staging the 64 KiB block of U
  const int sweep_threads = static_cast<int>(rows_per_block) * kSweepTeam < 256 ? 256 : static_cast<int>(rows_per_block) * kSweepTeam;
  p.rows_per_block = static_cast<int32_t>(rows_per_block);
  int64_t blocks = (R + rows_per_block - 1) / rows_per_block;
  if (blocks > 1024) blocks = 1024;
  const bool int_q = step->tensor.kind == DCQ_ROUND_INT;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DCQ_SWEEP(ZD)                                                                                       \
  do {                                                                                                      \
    auto kern = int_q ? gptq_sweep_kernel<ZD, true> : gptq_sweep_kernel<ZD, false>;                          \
    int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)), \
                       "hipFuncSetAttribute(sweep LDS)");                                                    \
    if (rc != DCQ_OK) return rc;                                                                             \
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(sweep_threads), lds, s, p);           \
  } while (0)
  switch (step->zero_domain) {
    case DCQ_ZERO_NONE:
      DCQ_SWEEP(DCQ_ZERO_NONE);
      break;
    case DCQ_ZERO_PRE_SCALE:
      DCQ_SWEEP(DCQ_ZERO_PRE_SCALE);
      break;
    default:
      DCQ_SWEEP(DCQ_ZERO_POST_SCALE);
      break;
  }
#undef DCQ_SWEEP
  DCQ_LAUNCH_CHECK("gptq_sweep_kernel launch");
  return DCQ_OK;
}

int64_t dcq_gptq_sweep_workspace_bytes(int64_t R, int64_t C, int32_t lazy_blocks) {
  if (R <= 0 || C <= 0 || lazy_blocks < 1) return 0;
  return static_cast<int64_t>(two_term_scratch_bytes(R, C, static_cast<int64_t>(kSweepCols) * lazy_blocks));
}

int dcq_gptq_sweep(const dcq_step* step, float* W, int64_t ldw, int64_t R, int64_t C, const float* U,
                   const float* scale, const float* zero, float zero_scalar, int32_t row_group_size,
                   int32_t col_group_size, int32_t col_groups, const int64_t* perm, float* Q, float* Err,
                   int32_t lazy_blocks, void* workspace, int64_t workspace_bytes, dcq_stream_t stream) {
  DCQ_REQUIRE(lazy_blocks >= 1 && lazy_blocks <= 64, DCQ_ERR_ARGUMENT, "lazy_blocks must be in [1, 64]");
  DCQ_REQUIRE(C > 0 && C < (1ll << 31), DCQ_ERR_ARGUMENT, "bad C");
  DCQ_REQUIRE(workspace == nullptr || ((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0 &&
                                       workspace_bytes >= dcq_gptq_sweep_workspace_bytes(R, C, lazy_blocks)),
              DCQ_ERR_ARGUMENT,
              "the workspace, when given, holds dcq_gptq_sweep_workspace_bytes(R, C, lazy_blocks) bytes, 256-byte aligned");
  if (R == 0) return DCQ_OK;
  const int64_t group_cols = static_cast<int64_t>(kSweepCols) * lazy_blocks;
  for (int64_t g0 = 0; g0 < C; g0 += group_cols) {
    const int64_t g_end = g0 + group_cols < C ? g0 + group_cols : C;
    for (int64_t c0 = g0; c0 < g_end; c0 += kSweepCols) {
      const int32_t bc = static_cast<int32_t>(C - c0 < kSweepCols ? C - c0 : kSweepCols);
      float* err_block = Err + (c0 - g0);
      int rc = dcq_gptq_sweep_block(step, W, ldw, R, C, U, static_cast<int32_t>(c0), bc, scale, zero, zero_scalar,
                                    row_group_size, col_group_size, col_groups, perm, Q, err_block, group_cols, stream);
      if (rc != DCQ_OK) return rc;
      const int64_t c_end = c0 + bc;
      if (c_end < g_end) {  // the rest of this group, right away (K = 128: a block short of 128 is the last one)
        rc = gemm_f32_on_path(err_block, group_cols, 0, U + c0 * C + c_end, C, W + c_end, ldw, R, g_end - c_end,
                              kSweepCols, -1.0f, 1.0f, static_cast<hipStream_t>(stream), kGemmFloat32);
        if (rc != DCQ_OK) return rc;
      }
    }
    if (g_end < C) {  // everything right of the group, once (a full group: its blocks are all 128 wide)
      // the path is chosen from the column count alone: a row's result must not depend on how many rows the call
      // holds (rows of one weight may be split over processes)
      const int path = C - g_end >= 2048 ? kGemmSplit : kGemmFloat32;
      const int rc = gemm_f32_on_path(Err, group_cols, 0, U + g0 * C + g_end, C, W + g_end, ldw, R, C - g_end, group_cols,
                                      -1.0f, 1.0f, static_cast<hipStream_t>(stream), path, static_cast<unsigned char*>(workspace));
      if (rc != DCQ_OK) return rc;
    }
  }
  return DCQ_OK;
}

}  // extern "C"
