// Fused range-ratio grid search (gfx950): every candidate ratio of every group in one pass over
// the weights.
//
// The reference evaluates one candidate at a time over the whole tensor and sums each group's error
// with the torch CPU backend, whose reduction is an 8-lane float32 SIMD loop.  To reproduce its
// result bit for bit the kernel mirrors that loop: a team of 8 GPU lanes owns one group and lane j
// holds exactly the elements SIMD lane j accumulates (element index = j mod 8), so the vector part
// of the sum happens in registers in the reference's order and only the final 8-way fold crosses
// lanes (DPP row shifts).  A wave therefore works on 8 groups; the weights are read from HBM once,
// transposed through a small per-wave LDS tile, and all K candidates are evaluated from registers.
//
// |Q(w) - w|^degree is evaluated in the weight's storage type.  For 16-bit storage the value is a
// pure function of the 15-bit magnitude pattern, so the block keeps the whole function as a 64 KiB
// table in LDS (built once per (dtype, degree) by dcq_pow_table) and the power costs one LDS read.
#include <string.h>

#include "dcq_common.cuh"

namespace dcq {

constexpr int kTeam = 8;                       // lanes per group = SIMD width of the mirrored reduction
constexpr int kGroupsPerWave = kWave / kTeam;  // 8
constexpr int kTableEntries = 32768;

enum PowMode { kPowTable = 0, kPowSquare = 1, kPowFloat = 2 };

// Threads per block by values per lane.  With the power table: 16 waves share one 64 KiB table per CU while the
// registers allow it (<= 128 VGPRs at 16 elements per lane); longer groups trade waves for registers and LDS.
// Without it nothing is shared, and four-wave workgroups (the same waves per SIMD) find room on a CU that other
// streams' kernels occupy in part, where a 16-wave workgroup waits for a CU to drain completely.
constexpr int search_block(int npl, int pow_mode) {
  return pow_mode != kPowTable ? 256 : (npl <= 16 ? 1024 : (npl == 32 ? 512 : 256));
}
constexpr int search_waves_per_simd(int npl) { return npl <= 16 ? 4 : (npl == 32 ? 2 : 1); }

struct SearchParams {
  dcq_step step;
  const void* w;
  int64_t G;
  int32_t S;
  const float* ratios;
  int32_t K;
  float degree;
  const uint16_t* table;
  float* best_max;
  float* best_min;
  float* best_ratio;
  float* best_error;
  float* errors;
  int32_t errors_unrounded;
};

// hardware float32 -> bfloat16 (round-to-nearest-even) used in the hot loop
__device__ __forceinline__ uint16_t f32_to_bf16_fast(float x) {
  const __bf16 h = static_cast<__bf16>(x);
  return __builtin_bit_cast(uint16_t, h);
}

template <int ST>
__device__ __forceinline__ uint16_t to_bits16(float v) {
  return ST == DCQ_BF16 ? f32_to_bf16_fast(v) : f32_to_f16(v);
}
template <int ST>
__device__ __forceinline__ float from_bits16(uint16_t b) {
  return ST == DCQ_BF16 ? bf16_to_f32(b) : f16_to_f32(b);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// two float32 -> two 16-bit patterns in one register (low half = first), round-to-nearest-even, hardware converts
template <int ST>
__device__ __forceinline__ uint32_t to_bits16x2(f32x2 v) {
  if (ST == DCQ_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <int ST>
__device__ __forceinline__ f32x2 from_bits16x2(uint32_t b) {
  if (ST == DCQ_BF16) return f32x2{__uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
  return __builtin_convertvector(__builtin_bit_cast(f16x2, b), f32x2);
}

// IEEE-exact division by a per-candidate scale with the reciprocal part hoisted: y = rcp(s) refined by one Newton
// step, q0 = x*y and two residual corrections -- the hardware's own division sequence without its range fix-ups,
// valid while 2^-60 <= s <= 2^60 and |x| <= 2^60 (checked per candidate; otherwise the plain division runs).
__device__ __forceinline__ float refined_rcp(float s) {
  const float y0 = __builtin_amdgcn_rcpf(s);
  return __builtin_fmaf(__builtin_fmaf(-s, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ f32x2 div_hoisted2(f32x2 x, float s, float y) {
  const f32x2 y2 = {y, y}, ns = {-s, -s};
  f32x2 q = x * y2;
  f32x2 r = __builtin_elementwise_fma(ns, q, x);
  q = __builtin_elementwise_fma(r, y2, q);
  r = __builtin_elementwise_fma(ns, q, x);
  return __builtin_elementwise_fma(r, y2, q);
}

// lane i of each 16-lane row receives lane i-1's value (row_shr:1)
__device__ __forceinline__ float row_shift_right1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
}

// sum over each team of 8 lanes, every lane gets it (quad swaps, then the mirror of each half row); the order of the
// additions is not the reference's: for bounds only
__device__ __forceinline__ float team8_sum_any_order(float v) {
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  return v;
}

template <int ST, int NPL, int ZD, int POW, bool INT_Q>
__global__ __launch_bounds__(search_block(NPL, POW), search_waves_per_simd(NPL)) void range_search_kernel(
    const SearchParams p) {
  constexpr int kSearchBlock = search_block(NPL, POW);
  constexpr int kGroupsPerBlock = kSearchBlock / kTeam;
  using St = Storage<ST>;
  using elem_t = typename St::elem_t;
  constexpr int EB = sizeof(elem_t);
  constexpr int S = NPL * kTeam;
  constexpr int kRowBytes = S * EB + 16;  // +16: consecutive groups start 4 banks apart
  constexpr int kPiecesPerGroup = S * EB / 16;
  constexpr bool PAIR = ST != DCQ_F32;  // a 16-element load is folded to 8 float lanes by one add
  constexpr int NV = PAIR ? NPL / 2 : NPL;
  constexpr int ROWS = NV / 4;
  static_assert(!PAIR || NPL % 2 == 0, "16-bit groups must be a multiple of 16 elements");

  // The power table is a static array: its LDS address is then a constant the compiler folds into the reads' offset
  // field (behind the dynamic array's base it costs one address add per read, a twentieth of the kernel's
  // instructions).  64 KiB is the most a static array may take; the rest is dynamic.
  __shared__ __align__(16) uint16_t table[POW == kPowTable ? kTableEntries : 8];
  extern __shared__ __align__(16) unsigned char smem[];
  float* cbs = reinterpret_cast<float*>(smem);  // [3][256] codebooks
  unsigned char* stage = smem + 3 * 256 * 4;

  const dcq_step& st = p.step;
  if (POW == kPowTable) {
    const uint4* src = reinterpret_cast<const uint4*>(p.table);
    uint4* dst = reinterpret_cast<uint4*>(table);
    for (int i = threadIdx.x; i < kTableEntries * 2 / 16; i += kSearchBlock) dst[i] = src[i];
  }
  {
    const dcq_rounding* r[3] = {&st.tensor, &st.scale, &st.zero};
    for (int k = 0; k < 3; ++k)
      if (r[k]->kind == DCQ_ROUND_CODEBOOK)
        for (int i = threadIdx.x; i < r[k]->codebook_size; i += kSearchBlock) cbs[k * 256 + i] = r[k]->codebook[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int j = lane & (kTeam - 1);  // mirrored SIMD lane
  const int gw = lane / kTeam;       // group within the wave
  unsigned char* wstage = stage + wave * (kGroupsPerWave * kRowBytes);
  const elem_t* w = static_cast<const elem_t*>(p.w);

  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kGroupsPerBlock; base < p.G;
       base += static_cast<int64_t>(gridDim.x) * kGroupsPerBlock) {
    const int64_t wbase = base + wave * kGroupsPerWave;  // first group of this wave
    const int64_t g = wbase + gw;
    // ---- HBM -> LDS, 16 bytes per lane, coalesced over the wave's 8 consecutive groups
    for (int piece = lane; piece < kGroupsPerWave * kPiecesPerGroup; piece += kWave) {
      const int gi = piece / kPiecesPerGroup, within = piece % kPiecesPerGroup;
      if (wbase + gi < p.G) {
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(w + (wbase + gi) * S) +
                                                        within * 16);
        *reinterpret_cast<uint4*>(wstage + gi * kRowBytes + within * 16) = v;
      }
    }
    // ---- LDS -> registers, lane j takes elements j, j+8, j+16, ... (same wave wrote them: no barrier needed)
    float x[NPL];
    const bool live = g < p.G;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const elem_t raw = *reinterpret_cast<const elem_t*>(wstage + gw * kRowBytes + (i * kTeam + j) * EB);
      x[i] = live ? St::load(raw) : 0.0f;
    }
    // ---- base range (DynamicRange.construct)
    float bmax, bmin = 0.0f;
    {
      float mx = ZD == DCQ_ZERO_NONE ? 0.0f : -INFINITY, mn = INFINITY;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        if (ZD == DCQ_ZERO_NONE) {
          mx = fmaxf(mx, fabsf(x[i]));
        } else {
          mx = fmaxf(mx, x[i]);
          mn = fminf(mn, x[i]);
        }
      }
      bmax = team_max(mx, kTeam);
      if (ZD != DCQ_ZERO_NONE) bmin = team_min(mn, kTeam);
      if (ZD == DCQ_ZERO_PRE_SCALE) {
        bmax = fmaxf(bmax, 0.0f);
        bmin = fminf(bmin, 0.0f);
      }
    }
    // packed integer path (16-bit weights): largest |x| of the group, for the guard of the hoisted division
    constexpr bool PACKED = INT_Q && ST != DCQ_F32 && POW != kPowFloat && NPL % 2 == 0;
    float gabs = 0.0f;
    if (PACKED) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) gabs = fmaxf(gabs, fabsf(x[i]));
    }
    float best_err = 0.0f;
    int best_k = 0;
    // Candidates that cannot win are dropped part-way (only the best candidate of a group is wanted unless the error
    // table is): every term of the sum is >= 0 and float addition is monotone, so the sum over a quarter / a half of
    // the elements, made a little smaller than any summation order could round it, is a lower bound of the candidate's
    // error; once that is at least one unit in the last place of the storage type above the group's best, the rounded
    // error cannot be smaller than the best (ties go to the earlier candidate anyway).  The wave moves on when all of
    // its 8 groups say so.  The winner and its error are those of the full evaluation.
    const bool prune = PACKED && p.errors == nullptr;
    // The scale, zero point and reciprocal of a candidate are the same for the 8 lanes of a group: lane j works them
    // out for candidate k0 + j, and the team then takes the candidates of the batch one by one from the lane that
    // holds them (LDS-crossbar permute), instead of every lane repeating the ~40 instructions for every candidate.
    const int team_first_lane4 = (lane & ~(kTeam - 1)) * 4;
    for (int k0 = 0; k0 < p.K; k0 += kTeam) {
      ScaleZero mine;
      float mine_y = 0.0f;
      int mine_ok = 0;
      {
        const int km = k0 + j < p.K ? k0 + j : p.K - 1;
        const float r = p.ratios[km];
        float cmax = bmax * r, cmin = 0.0f;
        if (ZD != DCQ_ZERO_NONE) cmin = bmin * r;
        if (ZD == DCQ_ZERO_PRE_SCALE) {
          cmax = fmaxf(cmax, 0.0f);
          cmin = fminf(cmin, 0.0f);
        }
        mine = scale_zero_from_range(st, cmax, cmin, cbs + 256, cbs + 512);
        if (PACKED) {
          mine_ok = mine.s >= 0x1p-60f && mine.s <= 0x1p60f && gabs + fabsf(mine.z) <= 0x1p60f;
          mine_y = refined_rcp(mine.s);
        }
      }
#pragma unroll 1
    for (int k = k0; k < k0 + kTeam && k < p.K; ++k) {
      const int from = team_first_lane4 + (k - k0) * 4;
      ScaleZero sz;
      sz.s = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mine.s)));
      sz.z = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mine.z)));
      float pw[NPL];
      bool done = false;
      if (PACKED) {
        const bool ok = __builtin_amdgcn_ds_bpermute(from, mine_ok) != 0;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) {  // uniform over the wave
          done = true;
          const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mine_y)));
          const f32x2 z2 = {sz.z, sz.z}, s2 = {sz.s, sz.s};
          float seen = 0.0f;  // sum of the terms so far, in any order
          bool dropped = false;
#pragma unroll
          for (int i = 0; i < NPL; i += 2) {
            if (NPL >= 8 && (i == NPL / 4 || i == NPL / 2) && prune && k > 0) {
#pragma unroll
              for (int m = (i == NPL / 4 ? 0 : NPL / 4); m < i; ++m) seen = seen + pw[m];
              const float lower = team8_sum_any_order(seen) * (1.0f - 0x1p-20f);
              const bool may_win = j == kTeam - 1 && lower < best_err * (1.0f + 0x1p-7f);
              if (__builtin_amdgcn_ballot_w64(may_win) == 0) {
                dropped = true;
                break;
              }
            }
            const f32x2 xv = {x[i], x[i + 1]};
            f32x2 t = ZD == DCQ_ZERO_POST_SCALE ? xv + z2 : xv;
            t = div_hoisted2(t, sz.s, y);
            if (ZD == DCQ_ZERO_PRE_SCALE) t = t + z2;
            // med3 == min(max(v, lo), hi) for the finite values the guard above admits: one instruction, not two
            const f32x2 q = {__builtin_amdgcn_fmed3f(rintf(t.x), st.tensor.lo, st.tensor.hi),
                             __builtin_amdgcn_fmed3f(rintf(t.y), st.tensor.lo, st.tensor.hi)};
            f32x2 d;
            if (ZD == DCQ_ZERO_PRE_SCALE) d = (q - z2) * s2;
            else if (ZD == DCQ_ZERO_POST_SCALE) d = q * s2 - z2;
            else d = q * s2;
            const f32x2 dq = from_bits16x2<ST>(to_bits16x2<ST>(d));          // Q(w) in the weight dtype
            const uint32_t eraw = to_bits16x2<ST>(dq - xv);                  // Q(w) - w in the weight dtype, two values
            if (POW == kPowTable) {
              // |.| and the table's two-byte stride in the same mask: byte offset = (bits & 0x7fff) * 2
              const unsigned char* tb = reinterpret_cast<const unsigned char*>(table);
              pw[i] = from_bits16<ST>(*reinterpret_cast<const uint16_t*>(tb + ((eraw << 1) & 0xfffeu)));
              pw[i + 1] = from_bits16<ST>(*reinterpret_cast<const uint16_t*>(tb + ((eraw >> 15) & 0xfffeu)));
            } else {
              const uint32_t eb = eraw & 0x7fff7fffu;                        // |Q(w) - w|
              const f32x2 a = from_bits16x2<ST>(eb);
              const f32x2 sq = from_bits16x2<ST>(to_bits16x2<ST>(a * a));
              pw[i] = sq.x;
              pw[i + 1] = sq.y;
            }
          }
          if (dropped) continue;
        }
      }
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        if (done) break;
        const float t = to_code_input<ZD>(x[i], sz.s, sz.z);
        const float q = INT_Q ? round_int(t, st.tensor.lo, st.tensor.hi) : apply_rounding(st.tensor, t, cbs);
        const float d = from_code<ZD>(q, sz.s, sz.z);
        if (ST == DCQ_F32) {
          const float a = fabsf(d - x[i]);
          pw[i] = POW == kPowSquare ? a * a : powf(a, p.degree);
        } else {
          const float dq = from_bits16<ST>(to_bits16<ST>(d));             // Q(w) in the weight dtype
          const uint16_t eb = to_bits16<ST>(dq - x[i]) & 0x7fffu;         // |Q(w) - w| in the weight dtype
          if (POW == kPowTable) {
            pw[i] = from_bits16<ST>(table[eb]);
          } else {
            const float a = from_bits16<ST>(eb);
            pw[i] = from_bits16<ST>(to_bits16<ST>(a * a));
          }
        }
      }
      // ---- the reference's vectorised inner sum, lane-local part
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const float v = PAIR ? pw[2 * m] + pw[2 * m + 1] : pw[m];
        const int slot = m < ROWS * 4 ? (m & 3) : 0;
        acc[slot] = acc[slot] + v;
      }
      const float part = ((acc[0] + acc[1]) + acc[2]) + acc[3];
      // ---- left-to-right fold over the 8 lanes: y_j = y_{j-1} + part_j
      float y = part;
#pragma unroll
      for (int t = 1; t < kTeam; ++t) {
        const float prev = row_shift_right1(y);
        y = j == 0 ? part : prev + part;
      }
      const float err = St::round(y);  // valid in lane j == 7
      if (p.errors && j == kTeam - 1 && live) p.errors[static_cast<int64_t>(k) * p.G + g] = p.errors_unrounded ? y : err;
      if (k == 0 || err < best_err) {
        best_err = err;
        best_k = k;
      }
    }
    }
    if (j == kTeam - 1 && live) {
      const float r = p.ratios[best_k];
      float cmax = bmax * r, cmin = 0.0f;
      if (ZD != DCQ_ZERO_NONE) cmin = bmin * r;
      if (ZD == DCQ_ZERO_PRE_SCALE) {
        cmax = fmaxf(cmax, 0.0f);
        cmin = fminf(cmin, 0.0f);
      }
      p.best_max[g] = cmax;
      if (ZD != DCQ_ZERO_NONE && p.best_min) p.best_min[g] = cmin;
      if (p.best_ratio) p.best_ratio[g] = r;
      if (p.best_error) p.best_error[g] = best_err;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pow table
// ---------------------------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(256) void pow_table_kernel(uint16_t* table, double degree, int special) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kTableEntries) return;
  const float a = Storage<ST>::load(static_cast<uint16_t>(i));
  float r;
  if (special == 2) {
    r = a * a;
  } else if (special == 3) {
    r = Storage<ST>::round(a * a) * a;
  } else {
    r = static_cast<float>(pow(static_cast<double>(a), degree));
  }
  table[i] = Storage<ST>::store(r);
}

// ---------------------------------------------------------------------------------------------
// Error of one given candidate: out[g] = sum_i pow(|a[g][i] - b[g][i]|, degree), the difference, magnitude and power
// each rounded to the storage type, accumulated in float32 in the order of the torch CPU backend's contiguous
// last-dimension sum (8 float lanes, four interleaved accumulators, 16-row cascade; 16-bit inputs folded pairwise)
// and left UNROUNDED so the caller can round it (per group) or add groups up first (per channel group / per tensor).
// ---------------------------------------------------------------------------------------------
struct GroupErrorParams {
  const void* a;
  const void* b;
  int64_t G, S;
  float degree;
  const uint16_t* table;
  int32_t pow_mode;
  int32_t accumulate;
  float* out;
};

// PRODUCTS: the operands are products x @ w in the storage type; their difference is rounded to that type, then
// widened to float32 and raised to the degree there with its sign (calib/search.py:803-810); plain float32 sum order.
template <int ST, bool PRODUCTS>
__global__ __launch_bounds__(256) void group_error_kernel(const GroupErrorParams p) {
  using St = Storage<ST>;
  using elem_t = typename St::elem_t;
  constexpr bool PAIR = ST != DCQ_F32 && !PRODUCTS;
  constexpr int kPerLoad = PAIR ? 2 * kTeam : kTeam;
  const int j = threadIdx.x & (kTeam - 1);
  const elem_t* a = static_cast<const elem_t*>(p.a);
  const elem_t* b = static_cast<const elem_t*>(p.b);
  const int64_t teams = static_cast<int64_t>(gridDim.x) * (256 / kTeam);
  // whole waves iterate together: the trip count depends on the wave's first group only
  for (int64_t g0 = static_cast<int64_t>(blockIdx.x) * (256 / kTeam); g0 < p.G; g0 += teams) {
    const int64_t g = g0 + threadIdx.x / kTeam;
    const bool live = g < p.G;
    const elem_t* ag = a + (live ? g : 0) * p.S;
    const elem_t* bg = b + (live ? g : 0) * p.S;
    auto term = [&](int64_t i) -> float {
      const float d = St::load(ag[i]) - St::load(bg[i]);
      if (PRODUCTS) {
        const float r = St::round(d);
        return p.pow_mode == kPowSquare ? r * r : powf(r, p.degree);
      }
      if (ST == DCQ_F32) {
        const float m = fabsf(d);
        return p.pow_mode == kPowSquare ? m * m : powf(m, p.degree);
      }
      const uint16_t eb = to_bits16<ST>(d) & 0x7fffu;
      if (p.pow_mode == kPowTable) return from_bits16<ST>(p.table[eb]);
      const float m = from_bits16<ST>(eb);
      return from_bits16<ST>(to_bits16<ST>(m * m));
    };
    auto load = [&](int64_t m) -> float {
      return PAIR ? term(m * kPerLoad + j) + term(m * kPerLoad + kTeam + j) : term(m * kPerLoad + j);
    };
    float tot = 0.0f;
    if (p.S < kPerLoad) {
      if (j == 0) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int64_t n4 = p.S / 4;
        for (int64_t i = 0; i < n4; ++i)
          for (int k = 0; k < 4; ++k) acc[k] = acc[k] + term(4 * i + k);
        for (int64_t i = n4 * 4; i < p.S; ++i) acc[0] = acc[0] + term(i);
        tot = ((acc[0] + acc[1]) + acc[2]) + acc[3];
      }
    } else {
      const int64_t nv = p.S / kPerLoad, rows = nv / 4;
      float acc[4][4];
#pragma unroll
      for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[l][k] = 0.0f;
      int64_t i = 0;
      while (i + 16 <= rows) {
        for (int t = 0; t < 16; ++t, ++i) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[0][k] = acc[0][k] + load(4 * i + k);
        }
        bool go = true;
#pragma unroll
        for (int l = 1; l < 4; ++l) {
          if (go) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              acc[l][k] = acc[l][k] + acc[l - 1][k];
              acc[l - 1][k] = 0.0f;
            }
            if ((i & (static_cast<int64_t>(15) << (4 * l))) != 0) go = false;
          }
        }
      }
      for (; i < rows; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = acc[0][k] + load(4 * i + k);
      }
#pragma unroll
      for (int l = 1; l < 4; ++l)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] = acc[0][k] + acc[l][k];
      for (int64_t m = rows * 4; m < nv; ++m) acc[0][0] = acc[0][0] + load(m);
      const float part = ((acc[0][0] + acc[0][1]) + acc[0][2]) + acc[0][3];
      if (j == 0)
        for (int64_t e = nv * kPerLoad; e < p.S; ++e) tot = tot + term(e);
#pragma unroll
      for (int t = 0; t < kTeam; ++t) tot = tot + __shfl(part, t, kTeam);  // meaningful in lane 0 (the tail's owner)
    }
    if (live && j == 0) p.out[g] = p.accumulate ? p.out[g] + tot : tot;
  }
}

double round_degree_to_storage(double degree, int storage) {
  // float -> storage -> double, on the host (bit manipulation; no fp16 hardware assumed)
  const float f = static_cast<float>(degree);
  uint32_t u;
  ::memcpy(&u, &f, 4);
  if (storage == DCQ_BF16) {
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float o;
    ::memcpy(&o, &u, 4);
    return o;
  }
  // float16: normal range only (degrees are small positive numbers)
  const int exp = static_cast<int>((u >> 23) & 0xff) - 127;
  if (exp < -14 || exp > 15) return f;
  const uint32_t keep = 13;  // 23 - 10 mantissa bits dropped
  u += (1u << (keep - 1)) - 1 + ((u >> keep) & 1u);
  u &= ~((1u << keep) - 1);
  float o;
  ::memcpy(&o, &u, 4);
  return o;
}

template <int ST, int NPL, int ZD, int POW>
static int launch_search(const SearchParams& p, hipStream_t stream) {
  constexpr int EB = sizeof(typename Storage<ST>::elem_t);
  constexpr int kSearchBlock = search_block(NPL, POW);
  constexpr int kGroupsPerBlock = kSearchBlock / kTeam;
  // dynamic part (the kernel's power table is static)
  const size_t lds = 3 * 256 * 4 + static_cast<size_t>(kSearchBlock / kWave) * kGroupsPerWave * (NPL * kTeam * EB + 16);
  int64_t blocks = (p.G + kGroupsPerBlock - 1) / kGroupsPerBlock;
  const int64_t most = 256 * (16 * kWave / kSearchBlock) * search_waves_per_simd(NPL) / 4;  // the chip's worth of waves
  if (blocks > most) blocks = most;
  const bool int_q = p.step.tensor.kind == DCQ_ROUND_INT;
  auto kern = int_q ? range_search_kernel<ST, NPL, ZD, POW, true> : range_search_kernel<ST, NPL, ZD, POW, false>;
  if (lds > 64 * 1024) {
    int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)),
                       "hipFuncSetAttribute(max dynamic LDS)");
    if (rc != DCQ_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kSearchBlock), lds, stream, p);
  DCQ_LAUNCH_CHECK("range_search_kernel launch");
  return DCQ_OK;
}

template <int ST, int NPL, int POW>
static int search_zero(const SearchParams& p, hipStream_t stream) {
  switch (p.step.zero_domain) {
    case DCQ_ZERO_NONE:
      return launch_search<ST, NPL, DCQ_ZERO_NONE, POW>(p, stream);
    case DCQ_ZERO_PRE_SCALE:
      return launch_search<ST, NPL, DCQ_ZERO_PRE_SCALE, POW>(p, stream);
    default:
      return launch_search<ST, NPL, DCQ_ZERO_POST_SCALE, POW>(p, stream);
  }
}

template <int ST, int POW>
static int search_size(const SearchParams& p, hipStream_t stream) {
  switch (p.S / kTeam) {
    case 2:
      return search_zero<ST, 2, POW>(p, stream);
    case 4:
      return search_zero<ST, 4, POW>(p, stream);
    case 8:
      return search_zero<ST, 8, POW>(p, stream);
    case 16:
      return search_zero<ST, 16, POW>(p, stream);
    case 32:
      return search_zero<ST, 32, POW>(p, stream);
    case 64:
      return search_zero<ST, 64, POW>(p, stream);
    default:
      set_error("range search: group size %d is not one of 16, 32, 64, 128, 256, 512", p.S);
      return DCQ_ERR_UNSUPPORTED;
  }
}

}  // namespace dcq

using namespace dcq;

extern "C" {

int dcq_pow_table(int32_t storage, double degree, uint16_t* table, dcq_stream_t stream) {
  DCQ_REQUIRE(table != nullptr, DCQ_ERR_ARGUMENT, "table is NULL");
  DCQ_REQUIRE(storage == DCQ_F16 || storage == DCQ_BF16, DCQ_ERR_ARGUMENT, "pow table exists for 16-bit storage only");
  const double d = round_degree_to_storage(degree, storage);
  // torch's bfloat16 pow special-cases 2 and 3 as repeated multiplication in bfloat16; float16 has no such case
  int special = 0;
  if (storage == DCQ_BF16 && d == 2.0) special = 2;
  if (storage == DCQ_BF16 && d == 3.0) special = 3;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (storage == DCQ_BF16) {
    hipLaunchKernelGGL(pow_table_kernel<DCQ_BF16>, dim3(kTableEntries / 256), dim3(256), 0, s, table, d, special);
  } else {
    hipLaunchKernelGGL(pow_table_kernel<DCQ_F16>, dim3(kTableEntries / 256), dim3(256), 0, s, table, d, special);
  }
  DCQ_LAUNCH_CHECK("pow_table_kernel launch");
  return DCQ_OK;
}

int dcq_range_search_groups(const dcq_step* step, const void* w, int32_t storage, int64_t G, int64_t S,
                            const float* ratios, int32_t K, double degree, const uint16_t* pow_table, float* best_max,
                            float* best_min, float* best_ratio, float* best_error, float* errors,
                            int32_t errors_unrounded, dcq_stream_t stream) {
  DCQ_REQUIRE(step != nullptr, DCQ_ERR_ARGUMENT, "step is NULL");
  DCQ_REQUIRE(step->tensor.kind == DCQ_ROUND_INT || step->tensor.kind == DCQ_ROUND_CODEBOOK, DCQ_ERR_UNSUPPORTED,
              "tensor dtype must be an integer or a codebook type");
  DCQ_REQUIRE(!(step->tensor.kind == DCQ_ROUND_CODEBOOK && step->zero_domain != DCQ_ZERO_NONE), DCQ_ERR_UNSUPPORTED,
              "floating-point dtype with a zero point is not implemented");
  DCQ_REQUIRE(G >= 0 && S > 0 && K > 0, DCQ_ERR_ARGUMENT, "bad sizes G=%lld S=%lld K=%d", (long long)G, (long long)S, K);
  if (G == 0) return DCQ_OK;
  DCQ_REQUIRE(w && ratios && best_max, DCQ_ERR_ARGUMENT, "NULL buffer");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15u) == 0, DCQ_ERR_ARGUMENT, "w must be 16-byte aligned");
  DCQ_REQUIRE(step->zero_domain == DCQ_ZERO_NONE || best_min, DCQ_ERR_ARGUMENT, "best_min is NULL");
  DCQ_REQUIRE(S <= 512 && S % (storage == DCQ_F32 ? 8 : 16) == 0, DCQ_ERR_UNSUPPORTED,
              "range search: group size %lld unsupported (multiple of 16, at most 512)", (long long)S);
  SearchParams p{};
  p.step = *step;
  p.w = w;
  p.G = G;
  p.S = static_cast<int32_t>(S);
  p.ratios = ratios;
  p.K = K;
  p.table = pow_table;
  p.best_max = best_max;
  p.best_min = best_min;
  p.best_ratio = best_ratio;
  p.best_error = best_error;
  p.errors = errors;
  p.errors_unrounded = errors_unrounded;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (storage == DCQ_F32) {
    p.degree = static_cast<float>(degree);
    return p.degree == 2.0f ? search_size<DCQ_F32, kPowSquare>(p, s) : search_size<DCQ_F32, kPowFloat>(p, s);
  }
  DCQ_REQUIRE(storage == DCQ_F16 || storage == DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage %d", storage);
  const double d = round_degree_to_storage(degree, storage);
  if (d == 2.0) {
    return storage == DCQ_BF16 ? search_size<DCQ_BF16, kPowSquare>(p, s) : search_size<DCQ_F16, kPowSquare>(p, s);
  }
  DCQ_REQUIRE(pow_table != nullptr, DCQ_ERR_ARGUMENT, "pow_table is NULL (build it with dcq_pow_table)");
  DCQ_REQUIRE((reinterpret_cast<uintptr_t>(pow_table) & 15u) == 0, DCQ_ERR_ARGUMENT, "pow_table must be 16-byte aligned");
  return storage == DCQ_BF16 ? search_size<DCQ_BF16, kPowTable>(p, s) : search_size<DCQ_F16, kPowTable>(p, s);
}

int dcq_group_error(const void* a, const void* b, int32_t storage, int64_t G, int64_t S, double degree,
                    const uint16_t* pow_table, float* out, dcq_stream_t stream) {
  DCQ_REQUIRE(G >= 0 && S > 0 && S <= (int64_t(1) << 24), DCQ_ERR_ARGUMENT, "bad sizes G=%lld S=%lld", (long long)G,
              (long long)S);
  if (G == 0) return DCQ_OK;
  DCQ_REQUIRE(a && b && out, DCQ_ERR_ARGUMENT, "NULL buffer");
  GroupErrorParams p{};
  p.a = a;
  p.b = b;
  p.G = G;
  p.S = S;
  p.out = out;
  p.table = pow_table;
  int64_t blocks = (G + 31) / 32;
  if (blocks > 65536) blocks = 65536;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (storage == DCQ_F32) {
    p.degree = static_cast<float>(degree);
    p.pow_mode = p.degree == 2.0f ? kPowSquare : kPowFloat;
    hipLaunchKernelGGL((group_error_kernel<DCQ_F32, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
  } else {
    DCQ_REQUIRE(storage == DCQ_F16 || storage == DCQ_BF16, DCQ_ERR_ARGUMENT, "bad storage %d", storage);
    p.pow_mode = round_degree_to_storage(degree, storage) == 2.0 ? kPowSquare : kPowTable;
    DCQ_REQUIRE(p.pow_mode != kPowTable || pow_table, DCQ_ERR_ARGUMENT, "pow_table is NULL (build it with dcq_pow_table)");
    if (storage == DCQ_BF16) {
      hipLaunchKernelGGL((group_error_kernel<DCQ_BF16, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
    } else {
      hipLaunchKernelGGL((group_error_kernel<DCQ_F16, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
    }
  }
  DCQ_LAUNCH_CHECK("group_error_kernel launch");
  return DCQ_OK;
}

int dcq_product_error(const void* y, const void* y0, int32_t storage, int64_t G, int64_t S, double degree,
                      int32_t accumulate, float* out, dcq_stream_t stream) {
  DCQ_REQUIRE(G >= 0 && S > 0 && S <= (int64_t(1) << 24), DCQ_ERR_ARGUMENT, "bad sizes G=%lld S=%lld", (long long)G,
              (long long)S);
  if (G == 0) return DCQ_OK;
  DCQ_REQUIRE(y && y0 && out, DCQ_ERR_ARGUMENT, "NULL buffer");
  GroupErrorParams p{};
  p.a = y;
  p.b = y0;
  p.G = G;
  p.S = S;
  p.out = out;
  p.accumulate = accumulate;
  p.degree = static_cast<float>(degree);
  p.pow_mode = p.degree == 2.0f ? kPowSquare : kPowFloat;
  int64_t blocks = (G + 31) / 32;
  if (blocks > 65536) blocks = 65536;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (storage) {
    case DCQ_F32:
      hipLaunchKernelGGL((group_error_kernel<DCQ_F32, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
      break;
    case DCQ_F16:
      hipLaunchKernelGGL((group_error_kernel<DCQ_F16, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
      break;
    case DCQ_BF16:
      hipLaunchKernelGGL((group_error_kernel<DCQ_BF16, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
      break;
    default:
      set_error("bad storage %d", storage);
      return DCQ_ERR_ARGUMENT;
  }
  DCQ_LAUNCH_CHECK("group_error_kernel (products) launch");
  return DCQ_OK;
}

}  // extern "C"
