/*
 * dcq.h -- C ABI of the MI355X (gfx950) calibration/quantization kernels.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream (passed as void*); nothing
 * here allocates device memory, synchronises the host, or knows about torch.  All functions return
 * DCQ_OK (0) or a negative dcq_status; dcq_last_error() gives the message of the calling thread's
 * last failure.  Arithmetic is IEEE float32 exactly as the reference's develop_dtype=float32 path
 * (true division, round-half-to-even, no fused multiply-add across reference ops).
 *
 * Layout convention ("canonical groups"): a tensor that the reference views as
 * (#g0, gs0, #g1, gs1, ...) is handed over as x[G][S] with every quantization group contiguous,
 * G = prod(#g_i), S = prod(gs_i).  For a 2-D weight with (1, g) groups this is the weight's own
 * memory; other group shapes are permuted by the caller.
 *
 * Each declaration cites the reference code (paths relative to the reference repository) whose
 * arithmetic it replaces.
 */
#ifndef DCQ_H_
#define DCQ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dcq_stream_t; /* hipStream_t */

enum dcq_status {
  DCQ_OK = 0,
  DCQ_ERR_ARGUMENT = -1,    /* null pointer, bad size, misaligned buffer */
  DCQ_ERR_UNSUPPORTED = -2, /* a combination the kernels do not implement (message says which) */
  DCQ_ERR_HIP = -3          /* a HIP runtime call failed */
};

/* storage type of a tensor in memory */
enum dcq_storage { DCQ_F32 = 0, DCQ_F16 = 1, DCQ_BF16 = 2, DCQ_F64 = 3 /* dcq_gemm_f64 operands and dcq_round_to_codebook only */ };

/* deepcompressor/data/zero.py:9-13 */
enum dcq_zero_domain { DCQ_ZERO_NONE = 0, DCQ_ZERO_PRE_SCALE = 1, DCQ_ZERO_POST_SCALE = 2 };

/* How one value is snapped onto a data type: one call of simple_quantize
 * (deepcompressor/quantizer/impl/simple.py:13-55). */
enum dcq_round_kind {
  DCQ_ROUND_F32 = 0,      /* torch.float32 scale dtype: identity                                     */
  DCQ_ROUND_F16 = 1,      /* .to(float16).to(float32)                                  (simple.py:23-30) */
  DCQ_ROUND_BF16 = 2,     /* .to(bfloat16).to(float32)                                                 */
  DCQ_ROUND_INT = 3,      /* round half-to-even, clamp to [lo, hi]                      (simple.py:44-51) */
  DCQ_ROUND_CODEBOOK = 4, /* clamp to [lo, hi], nearest codebook value, tie -> upper    (simple.py:37-43,
                             csrc/quantize/quantize.cu:69-99)                                          */
  DCQ_ROUND_EXP = 5       /* exp2(clamp(floor(log2 x), lo, hi))                         (simple.py:32-36) */
};

typedef struct dcq_rounding {
  int32_t kind;          /* dcq_round_kind */
  float lo, hi;          /* see dcq_round_kind */
  const float* codebook; /* device pointer, ascending, codebook_size <= 256 (DCQ_ROUND_CODEBOOK only) */
  int32_t codebook_size;
  int32_t has_clamp;     /* optional extra clamp after rounding (the reference's quant_range for fp dtypes) */
  float clamp_lo, clamp_hi;
} dcq_rounding;

/* One quantization step with a single-level group scale: the per-group arithmetic of
 * QuantScaleInfo.quantize (deepcompressor/quantizer/impl/scale.py:142-242), rtn_quantize
 * (quantizer/kernel/rtn.py:103-117) and the dequantization of QuantizerImpl._quantize
 * (quantizer/impl/base.py:262-269). */
typedef struct dcq_step {
  dcq_rounding tensor; /* the quantization dtype with its (intersected) quant range */
  dcq_rounding scale;  /* dtype of the group scale */
  dcq_rounding zero;   /* dtype of the zero point (tensor dtype for PreScale, scale dtype for PostScale) */
  int32_t zero_domain; /* dcq_zero_domain */
  float q_span;        /* linear_tensor_quant_span: qmax - qmin with zero point, else qmax (scale.py:116-123) */
  float q_min;         /* tensor_quant_range.min, used by the zero-point formula (scale.py:229-236) */
  int32_t has_bound_min, has_bound_max; /* RangeBound (data/range.py:257-275) */
  float bound_min, bound_max;
} dcq_step;

const char* dcq_version(void);
const char* dcq_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Codebook rounding.  Replaces _C.round_to_nearest_in_codebook_cuda(tensor, codebook, inplace, bnb)
 * (deepcompressor/csrc/pybind.cpp:6-10, csrc/quantize/quantize.cu:104-160) called from
 * Codebook.round (data/codebook.py:37-50).  y may alias x.  x, y: n values of `storage` type.
 * `codebook`: the table's values, ascending, float32 -- or float64 when `storage` is DCQ_F64: the reference
 * instantiates its kernel for float and double tensors (quantize.cu:145) with the table in the tensor's type, and
 * compares in that type.  DCQ_F16 / DCQ_BF16 tensors (no counterpart in the reference, whose caller casts to the
 * table's type first) are widened to float32 for the comparison.
 * The reference's `bnb` variant is never selected by its own callers and is not provided.
 * ------------------------------------------------------------------------------------------- */
int dcq_round_to_codebook(const void* x, void* y, int64_t n, int32_t storage, const void* codebook,
                          int32_t codebook_size, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused fake quantization of x[G][S], one scale (and zero) per group: range measurement ->
 * scale / zero -> round-to-nearest codes -> dequantization, one read and one write of the tensor.
 * Replaces the chain DynamicRange.measure (data/range.py:277-356) -> QuantScaleInfo.quantize
 * (quantizer/impl/scale.py:142-242) -> rtn_quantize (quantizer/kernel/rtn.py:68-117) -> dequant
 * (quantizer/impl/base.py:262-279) that Quantizer.quantize (quantizer/processor.py:93-183) runs for
 * a single-step, single-scale-level config.
 *
 *   range_max/range_min : optional static per-group range [G] (DynamicRange with max set, ratio None);
 *                         NULL -> measured from x.  range_min is ignored without a zero point.
 *   has_ratio/ratio     : DynamicRange.ratio applied to the measured range.
 *   dequant             : optional output, `storage` type, same layout as x.
 *   codes               : optional output, float32 integer/codebook values (the reference's qdata).
 *   scale, zero         : optional outputs [G] float32 (zero is all 0 without a zero point).
 *   nonfinite           : optional device int32, |= 4 (DCQ_NONFINITE_SEEN) when an element of x is NaN or +-Inf --
 *                         the reference's `assert not tensor.isnan().any()` / `.isinf().any()` on every call
 *                         (quantizer/impl/base.py:250-251), detected while the tensor is read anyway: the codes of
 *                         these kernels are clamped with min / max instructions and cannot show a NaN themselves.
 *                         The caller zeroes it and reads it when it wants to know.  Every entry point below that
 *                         takes `nonfinite` treats it this way.
 * ------------------------------------------------------------------------------------------- */
#define DCQ_NONFINITE_SEEN 4
int dcq_fake_quant_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                          const float* range_max, const float* range_min, int32_t has_ratio, float ratio,
                          void* dequant, float* codes, float* scale, float* zero, int32_t* nonfinite,
                          dcq_stream_t stream);

/* dcq_fake_quant_groups_outer: the LAST step of a progressive (two-step) quantizer in one pass, e.g. the uint4 g128
 * step of QoQ on top of its per-channel sint8 step (quantizer/impl/base.py:211-279): the group's data is
 * clamp(x / outer[g], outer_lo, outer_hi) (base.py:221-224, not rounded), measured, quantized with this step's scale
 * and zero (any scale / zero dtype with one level, e.g. sint8 integer scales), dequantized, clamped to
 * [deq_lo, deq_hi] when deq_clamp and multiplied back by outer[g] (base.py:270-278), written as `out_storage`.
 * Other arguments as dcq_fake_quant_groups. */
int dcq_fake_quant_groups_outer(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                                const float* outer, float outer_lo, float outer_hi, int32_t deq_clamp, float deq_lo,
                                float deq_hi, const float* range_max, const float* range_min, int32_t has_ratio,
                                float ratio, void* dequant, int32_t out_storage, float* codes, float* scale, float* zero,
                                int32_t* nonfinite, dcq_stream_t stream);

/* dcq_fake_quant_progressive: BOTH steps of a two-step progressive quantizer in one pass over x[rows][row_len] when
 * the first step has one group per row, no zero point and a float scale (QoQ W4: per-channel sint8 with fp16 scales,
 * then uint4 groups of S with integer scales; quantizer/impl/base.py:211-279, examples/llm/configs/qoq-g128.yaml):
 *     first_scale[r] = Q(|max|x[r]| * first_ratio / first->q_span|)   (0 -> 1)
 *     the last step on clamp(x / first_scale[r], outer_lo, outer_hi) as dcq_fake_quant_groups_outer does.
 * Limits (DCQ_ERR_UNSUPPORTED beyond them; use dcq_group_range + dcq_range_to_scale + dcq_fake_quant_groups_outer):
 * rows of at most 2048 16-byte pieces, S a power-of-two number (<= 64) of 16-byte pieces dividing row_len, the last
 * step an integer type with one scale level.  first_scale [rows], scale / zero [rows * row_len / S]. */
int dcq_fake_quant_progressive(const dcq_step* first, const dcq_step* last, const void* x, int32_t storage,
                               int64_t rows, int64_t row_len, int64_t S, int32_t first_has_ratio, float first_ratio,
                               int32_t has_ratio, float ratio, float outer_lo, float outer_hi, int32_t deq_clamp,
                               float deq_lo, float deq_hi, void* dequant, float* codes, float* first_scale,
                               float* scale, float* zero, int32_t* nonfinite, dcq_stream_t stream);

/* dcq_fake_quant_groups_two_level: fake quantization with TWO scale levels, the first one per tensor -- NVFP4:
 * float4 codes in groups of 16 with float8 group scales relative to one 16-bit scale of the whole tensor
 * (quantize_scale, quantizer/impl/scale.py:20-64; examples/diffusion/configs/svdquant/nvfp4.yaml).  A reduction
 * kernel finds max |x| of the tensor (into *tensor_absmax, caller-provided device float), then one pass computes
 *     level0 = Q0(|max * ratio / q_span| / level0_span),   level1[g] = Q1(|group max * ratio / q_span| / level0),
 *     scale[g] = level1[g] * level0 (zero -> 1),   codes, dequantized values
 * with Q0 = `level0` (a float type), Q1 = step->scale (e.g. the float8 codebook).  No zero point.  Outputs are optional. */
int dcq_fake_quant_groups_two_level(const dcq_step* step, const dcq_rounding* level0, float level0_span, const void* x,
                                    int32_t storage, int64_t G, int64_t S, int32_t has_ratio, float ratio,
                                    float* tensor_absmax, void* dequant, float* codes, float* level0_out,
                                    float* level1_out, float* scale, int32_t* nonfinite, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Building blocks of the general path (multi-level scales, progressive two-step configs, custom
 * quantization kernels).  Same references as above, one reference op chain per function.
 * ------------------------------------------------------------------------------------------- */

/* Per-group range of x[G][S]: max|x| (zero_domain NONE) or (max, min).  When `outer` is given the
 * measured value is clamp(x / outer[g], outer_lo, outer_hi), i.e. the un-rounded output of a previous
 * progressive step (quantizer/impl/base.py:221-224).  data/range.py:309-322. */
int dcq_group_range(const void* x, int32_t storage, int64_t G, int64_t S, int32_t zero_domain,
                    const float* outer, float outer_lo, float outer_hi, float* vmax, float* vmin,
                    int32_t* nonfinite, dcq_stream_t stream);

/* ratio, PreScale clamp, static bound, RangeBound, span / q_span -> un-quantized linear scale [n];
 * vmin is updated in place to the range minimum the zero point must use.  `vavg` (NULL, or the group
 * means [n]): the range of a floating-point type with a zero point, symmetric about the mean with
 * half-width max(vmax - vavg, vavg - vmin) * ratio.  data/range.py:323-355, quantizer/impl/scale.py:173-188. */
int dcq_range_to_scale(const dcq_step* step, int64_t n, const float* vmax, float* vmin, const float* vavg,
                       int32_t has_ratio, float ratio, const float* static_max, const float* static_min,
                       float* linear_scale, dcq_stream_t stream);

/* One level of quantize_scale (quantizer/impl/scale.py:46-62) on s[P][Q] (Q children per parent):
 *   reduce != 0: level[p] = Q(max_q s[p][q] / span), s[p][q] /= level[p]   (non-last level, or last level with Q > 1)
 *   reduce == 0: level[i] = Q(s[i])                                        (last level, one scale per group)
 * s must hold non-negative values (the caller passes |linear scale|; this function takes fabsf first). */
int dcq_scale_level(const dcq_rounding* rounding, float* s, int64_t P, int64_t Q, int32_t reduce, float span,
                    float* level, dcq_stream_t stream);

/* out[i] = Q(in[i]) for values of any sign: simple_quantize (quantizer/impl/simple.py:13-55) on a float32 array. */
int dcq_round_values(const dcq_rounding* rounding, const float* in, float* out, int64_t n, dcq_stream_t stream);

/* compound[i] = a[i] * b[i / Q] with zeros replaced by 1 when fix_zero (data/scale.py:66-69,91-117). */
int dcq_scale_join(const float* a, const float* b, int64_t n, int64_t Q, int32_t fix_zero, float* compound,
                   dcq_stream_t stream);

/* zero[i] = Q(q_min - vmin[i] / scale[i]) (PreScale) or Q(q_min * scale[i] - vmin[i]) (PostScale).
 * quantizer/impl/scale.py:229-236. */
int dcq_zero_point(const dcq_step* step, int64_t n, const float* vmin, const float* scale, float* zero,
                   dcq_stream_t stream);

/* rtn_quantize (quantizer/kernel/rtn.py:103-117) + dequantization (quantizer/impl/base.py:262-279) of
 * x[G][S] with given per-group scale / zero.  `outer` as in dcq_group_range; with it the dequantized
 * value is clamped to [deq_lo, deq_hi] when deq_clamp (needs_dequant_saturation) and multiplied back.
 * x may be float32 working data (storage DCQ_F32) while dequant is written as `out_storage`.
 * `round_delta` (NULL, or [G][S] float32): codes = clamp(floor(t) + delta) instead of round-to-nearest
 * (quantizer/impl/simple.py:47-51), integer types only. */
int dcq_quantize_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                        const float* scale, const float* zero, const float* outer, float outer_lo, float outer_hi,
                        int32_t deq_clamp, float deq_lo, float deq_hi, void* dequant, int32_t out_storage,
                        float* codes, const float* round_delta, int32_t* nonfinite, dcq_stream_t stream);

/* out[g][i] = clamp(x[g][i] / outer[g], lo, hi) in float32: the working tensor a progressive step hands to the next
 * one (quantizer/impl/base.py:221-224); only needed when a caller-supplied kernel object must see that tensor. */
int dcq_prescale_groups(const void* x, int32_t storage, int64_t G, int64_t S, const float* outer, float lo, float hi,
                        float* out, dcq_stream_t stream);

/* Dequantize given codes[G][S] (float32), e.g. those returned by a GPTQ kernel: same arithmetic as above. */
int dcq_dequantize_groups(const dcq_step* step, const float* codes, int64_t G, int64_t S, const float* scale,
                          const float* zero, const float* outer, int32_t deq_clamp, float deq_lo, float deq_hi,
                          void* dequant, int32_t out_storage, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Range-ratio grid search, all candidates per group in one pass over the weights.
 * Replaces the ask/tell loop of DynamicRangeCalibrator (deepcompressor/calib/range.py:73-221) as
 * driven by SearchBasedCalibrator._calibrate_wgts with the TensorError objective and Group
 * granularity (calib/search.py:776-790), clamp-based (static weights, calib/range.py:69-71).
 *
 * dcq_pow_table: table[i] = storage(pow(value_of_bits(i), degree)) for the 32768 non-negative 16-bit
 * patterns of `storage` (F16 or BF16), reproducing Tensor.pow_(degree) of the torch CPU backend on
 * that dtype (exponent rounded to the dtype first; bf16 special cases 2 and 3).  calib/search.py:781.
 *
 * dcq_range_search_groups: for each group g of w[G][S] and each ratio k (in the given order, the
 * reference's is [1.0, then descending]): candidate range = base range * ratio (data/range.py:358-395),
 * fake-quantize with that static range, error = sum over the group of |Q(w) - w|^degree evaluated in
 * the weight's storage type in the summation order of the torch CPU backend, keep the first strictly
 * smallest.  Outputs [G]: best_max, best_min (NULL without zero point), best_ratio (float32 of the
 * ratio), best_error (storage type, as float32 values); errors (optional, [K][G] float32 values: rounded to
 * the storage type as the reference's per-group error is, or, with errors_unrounded, the float32 sums before
 * that rounding, for callers that add groups up -- ChannelGroup / Layer granularity, calib/search.py:783-787).
 * S must be a multiple of 16 (16-bit storage) or 8 (float32) and at most 512.
 *
 * dcq_group_error: the error of ONE given candidate, for searches the fused kernel does not cover (progressive
 * steps, multi-level scales, groups longer than 512): out[g] = sum_i pow(|a[g][i] - b[g][i]|, degree) over
 * contiguous groups, difference / magnitude / power rounded to the storage type and summed in float32 in the order
 * of the torch CPU backend's contiguous last-dimension sum, left unrounded (calib/search.py:779-787).
 * ------------------------------------------------------------------------------------------- */
int dcq_pow_table(int32_t storage, double degree, uint16_t* table, dcq_stream_t stream);

int dcq_range_search_groups(const dcq_step* step, const void* w, int32_t storage, int64_t G, int64_t S,
                            const float* ratios, int32_t K, double degree, const uint16_t* pow_table,
                            float* best_max, float* best_min, float* best_ratio, float* best_error, float* errors,
                            int32_t errors_unrounded, dcq_stream_t stream);

int dcq_group_error(const void* a, const void* b, int32_t storage, int64_t G, int64_t S, double degree,
                    const uint16_t* pow_table, float* out, dcq_stream_t stream);

/* Error of products / outputs for the ProductsError and OutputsError objectives (calib/search.py:803-810, 825-830):
 * out[g] (+)= sum_i pow(float32(storage(y[g][i] - y0[g][i])), degree) -- the difference rounded to the operands'
 * storage type, the power (with the difference's sign, as the reference has no abs here) and the sum in float32.
 * `accumulate` adds to out, which is how the reference adds batches up. */
int dcq_product_error(const void* y, const void* y0, int32_t storage, int64_t G, int64_t S, double degree,
                      int32_t accumulate, float* out, dcq_stream_t stream);

/* Self-test hook (no reference counterpart): evaluates the kernels' division-by-group-scale routine
 * (fast[i]; used_fast[i] says whether the reciprocal path or the plain division was taken) next to the
 * hardware's IEEE division (ref[i]) for n pairs (x[i], s[i]), so a test can assert bitwise equality.
 * one_step (optional): the one-correction form the kernels use for divisors of at most 16 significant bits. */
int dcq_selftest_division(const float* x, const float* s, float* fast, float* ref, int32_t* used_fast,
                          float* one_step, int64_t n, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GPTQ (deepcompressor/quantizer/kernel/gptq.py:128-285).
 *
 * dcq_hessian_accumulate: H[C][C] += alpha * x^T x for x[T][C] (float16 / bfloat16, row-major), float32
 * accumulation on the matrix cores; replaces `hessian += matmul(x.t(), x)` with x pre-multiplied by
 * sqrt(2 / num_samples) (gptq.py:184-194): pass alpha = 2 / num_samples.  Only the upper triangle is
 * computed and mirrored; the result does not depend on launch timing (no atomics).  `workspace` must hold
 * dcq_hessian_workspace_bytes(T, C) bytes (256-byte aligned): a zero-padded transpose of x plus split-K
 * partial tiles (16-byte aligned x with C a multiple of 256, at most 8192, and T a multiple of 64 is read as it is;
 * the space for the transpose is then left unused).
 *
 * dcq_gptq_sweep_block: one block (<= 128 columns, the reference's block_size) of the column loop
 * (gptq.py:236-268).  W[R][ldw] is the float32 working weight with columns already permuted by `perm`
 * (act-order) and all previous blocks' lazy updates applied; U[C][C] is the upper Cholesky factor of
 * H^-1 in permuted order.  For each column c of the block and each row: clamp to the step's RangeBound,
 * quantize with the scale/zero of original column group perm[c] / col_group_size (gptq.py:246-259),
 * error = (w - dequant) / U[c][c], and W[:, c+1 : c0+bc] -= error * U[c][c+1 : c0+bc].  Outputs the codes
 * Q[R][ldw] (permuted column order, block columns only) and Err[R][ld_err] (128 columns written, zeros past bc)
 * for the caller's lazy update W[:, c0+bc:] -= Err @ U[c0:c0+bc, c0+bc:] (gptq.py:269); ld_err > 128 lets the
 * caller collect the errors of several blocks side by side and update far columns with one larger product.
 * scale/zero: [R / row_group_size][col_groups].
 *
 * dcq_gptq_sweep: the whole column loop (gptq.py:234-269) in one call -- every block through
 * dcq_gptq_sweep_block and the lazy updates through dcq_gemm_f32, all enqueued on `stream` without returning
 * to the host.  The updates are regrouped: the errors of `lazy_blocks` consecutive blocks are kept side by
 * side in Err[R][128 * lazy_blocks]; columns inside such a group are updated after every block, the columns
 * right of it once per group with one product of depth 128 * lazy_blocks (the same sums as the reference's
 * per-block update; only the float32 summation order differs).  U is [C][C] with leading dimension C.
 * `workspace`: NULL, or dcq_gptq_sweep_workspace_bytes(R, C, lazy_blocks) bytes (256-byte aligned) of scratch, with which the far updates of 2048 or more
 * columns run as dcq_gemm_f32_two_term products (half the matrix instructions of the three-term path they take
 * otherwise, float32-GEMM accuracy either way; a row's result depends on the choice but never on R).
 * ------------------------------------------------------------------------------------------- */
int64_t dcq_hessian_workspace_bytes(int64_t T, int64_t C);

int dcq_hessian_accumulate(const void* x, int32_t storage, int64_t T, int64_t C, float alpha, float* H, void* workspace,
                           int64_t workspace_bytes, dcq_stream_t stream);

/* dcq_hessian_accumulate with its two optional controls as arguments (neither has a reference counterpart, neither
 * changes the result):
 *   round_width  : scheduling hint -- launch the matrix-core kernel in rounds of at most this many workgroups (rounded
 *                  down to a multiple of 8, one per XCD in turn) instead of one per compute unit, which leaves the
 *                  other compute units to kernels of other streams: a Hessian that runs beside latency-bound
 *                  single-workgroup kernels otherwise holds every compute unit until a workgroup of its round ends.
 *                  0: one workgroup per compute unit.
 *   start_event, stop_event : measurement -- hipEvent_t handles (created by the caller with timing enabled) recorded
 *                  on `stream` right before and right after the matrix-core kernel launches (one per round); NULL: none. */
int dcq_hessian_accumulate_ex(const void* x, int32_t storage, int64_t T, int64_t C, float alpha, float* H,
                              void* workspace, int64_t workspace_bytes, int32_t round_width, void* start_event,
                              void* stop_event, dcq_stream_t stream);

int dcq_gptq_sweep_block(const dcq_step* step, float* W, int64_t ldw, int64_t R, int64_t C, const float* U, int32_t c0,
                         int32_t bc, const float* scale, const float* zero, float zero_scalar, int32_t row_group_size,
                         int32_t col_group_size, int32_t col_groups, const int64_t* perm, float* Q, float* Err,
                         int64_t ld_err, dcq_stream_t stream);

/* *flags |= 1 when x[0 .. n) (float32, 16-byte aligned) holds a NaN, |= 2 when it holds an infinity; the caller zeroes
 * *flags first and reads it after the stream has run.  One read of x.  Replaces the reference's
 * `assert not t.isnan().any()` / `assert not t.isinf().any()` pairs on codes and inverse factors
 * (quantizer/impl/base.py:250-251, quantizer/kernel/gptq.py:229-230, 272). */
int dcq_nonfinite_flags(const float* x, int64_t n, int32_t* flags, dcq_stream_t stream);

/* dst[r][c] = zero_columns[index[c]] ? 0 : float(src[r][index[c]]) for a row-major source of `storage` type (DCQ_F32 /
 * DCQ_F16 / DCQ_BF16) and a row-major float32 destination (leading dimensions ld_src, ld_dst in elements;
 * `zero_columns` [C] bytes indexed by SOURCE column, or NULL; src != dst; 16-byte aligned pointers, leading
 * dimensions multiples of 4, C <= 36864).  Replaces the act-order gathers around the sweep, `W[:, perm]` with the dead
 * channels' columns zeroed (gptq.py:199-208) and `Q[:, inverse_perm]` (gptq.py:273-275): each row is staged in LDS
 * and gathered there. */
int dcq_gather_columns(const void* src, int32_t storage, int64_t ld_src, int64_t R, int64_t C, const int64_t* index,
                       const uint8_t* zero_columns, float* dst, int64_t ld_dst, dcq_stream_t stream);

int64_t dcq_gptq_sweep_workspace_bytes(int64_t R, int64_t C, int32_t lazy_blocks);

int dcq_gptq_sweep(const dcq_step* step, float* W, int64_t ldw, int64_t R, int64_t C, const float* U,
                   const float* scale, const float* zero, float zero_scalar, int32_t row_group_size,
                   int32_t col_group_size, int32_t col_groups, const int64_t* perm, float* Q, float* Err,
                   int32_t lazy_blocks, void* workspace, int64_t workspace_bytes, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense float32 linear algebra of GPTQ on the matrix cores.
 *
 * dcq_gemm_f32: C[M][N] = beta * C + alpha * A * B with B row-major [K][N] (ldb) and A either row-major
 * [M][K] (a_kmajor = 0) or stored transposed [K][M] (a_kmajor = 1), leading dimension lda.  K must be a
 * multiple of 32; 16-byte aligned A, B with lda, ldb multiples of 4 take the fast load path.  Replaces the lazy update
 * `W[:, c_end:] -= Err @ U[c0:c_end, c_end:]` (gptq.py:269) and is the building block of the routine below.
 * Accuracy is that of a float32 GEMM for every size (error within a few 2^-24 of sum |a||b| per element); products of
 * 128 or more 128 x 128 tiles are computed on the bf16 matrix pipe from exact three-term bfloat16 splits of the
 * operands, so their low bits differ from those of a float32 multiply-accumulate chain.
 *
 * dcq_gemm_f32_two_term: the same product from two float16 terms per operand, for any size.  Each row of A and each
 * column of B is first scaled by the power of two that brings its largest magnitude into [2^14, 2^15) (the scales
 * are taken out of the result again, exactly), then every value is split into h = float16(v) and l = float16(v - h)
 * (once, into scratch planes laid out for the product kernel) and the product accumulates h h' + h l' + l h' in
 * float32: three float16 matrix instructions where the three-term path needs six.  A term pair misses at most 2^-22
 * of its value or 2^-39 of the largest magnitude in its row or column, whichever is larger, and the term left out is
 * of that size too, so the error per element stays within 3 * 2^-22 of sum |a||b| however short K is (measured:
 * 4.7e-7 at K = 32; 1.4e-7 ... 2e-7 from K = 512 on, where a float32 multiply-accumulate chain has 2.5e-7 ...
 * 3.6e-7, also with rows and columns graded over twelve decades).  A row or column whose largest magnitude is zero,
 * denormal or not finite is left unscaled (non-finite values give non-finite results).  Requires 16-byte aligned A
 * and B, lda and ldb multiples of 4, K a multiple of 32, N (and M when a_kmajor) a multiple of 4
 * (DCQ_ERR_UNSUPPORTED otherwise); `workspace`: dcq_gemm_f32_two_term_workspace_bytes(M, N, K) bytes, 256-byte
 * aligned, for the magnitudes and for the operands' term planes (about the size of the operands, padded to whole
 * 128-wide tiles).  No reference counterpart beyond the float32 matmul it stands for; the products of 128 or more
 * tiles of dcq_cholesky_inverse_upper and the far updates of dcq_gptq_sweep run on it.
 *
 * dcq_cholesky_inverse_upper: U[n][n] (upper triangular, row-major) with U^T U = S^-1 where
 *     S = P H' P^T + diag_add * I,   H' = H with zero diagonal entries set to 1 (gptq.py:199-201),
 *     P = row/column gather by perm[n] (act-order, gptq.py:204-208; NULL = identity).
 * Replaces `cholesky -> cholesky_inverse -> cholesky(upper=True)` (gptq.py:215-233) by one blocked
 * factorisation of the index-reversed matrix plus a blocked triangular inverse, which yields the same
 * factor.  H is not modified.  *info (device int32) receives 0, or the 1-based index (in the reversed
 * order) of the first non-positive pivot -- the caller then raises diag_add and retries, as the
 * reference does on torch's LinAlgError.  `workspace`: dcq_cholesky_workspace_bytes(n) bytes, 256-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int dcq_gemm_f32(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, float alpha, float beta, dcq_stream_t stream);

int64_t dcq_gemm_f32_two_term_workspace_bytes(int64_t M, int64_t N, int64_t K);

int dcq_gemm_f32_two_term(const float* A, int64_t lda, int32_t a_kmajor, const float* B, int64_t ldb, float* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, float alpha, float beta, void* workspace, int64_t workspace_bytes,
                          dcq_stream_t stream);

int64_t dcq_cholesky_workspace_bytes(int64_t n);

int dcq_cholesky_inverse_upper(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U, void* workspace,
                               int64_t workspace_bytes, int32_t* info, dcq_stream_t stream);

/* dcq_cholesky_inverse_upper_static: the same routine for callers that cannot allocate device memory or pass a
 * workspace -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  The workspace is a 41.25 MiB array built into the library (one copy per GPU, held until
 * the process ends), so n is limited to 2048 (DCQ_ERR_UNSUPPORTED beyond); nothing is allocated or freed.  Same
 * kernels and results as dcq_cholesky_inverse_upper, bit for bit.  All calls on one GPU share that array: calls that
 * overlap on one GPU (different streams or threads) race and must be serialised by the caller.  The first call on a
 * device does one-time host-side set-up (the array's address, kernel attributes): make one eager call per device
 * before capturing calls into a HIP graph. */
int dcq_cholesky_inverse_upper_static(const float* H, int64_t n, const int64_t* perm, float diag_add, float* U,
                                      int32_t* info, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Rotation (deepcompressor/calib/rotate.py:66-133, utils/math/hadamard.py:48-61).
 *
 * dcq_gemm_f64: C_b[M][N] = A_b * B_b for b < batch, accumulated in float64 on the matrix cores.  Every operand
 * has its own storage type (DCQ_F64 / DCQ_F32 / DCQ_F16 / DCQ_BF16): inputs are widened exactly, the result is
 * rounded double -> float -> storage, which is what `Tensor.to(dtype)` does to a float64 tensor.  B is
 * row-major [K][N]; A is row-major [M][K] (a_kmajor = 0) or stored transposed [K][M] (a_kmajor = 1).  Batch
 * strides are in elements (0 = shared operand).  One call replaces
 *   `matmul(weight.view(-1, n).to(float64), rotation).to(dtype)`                  rotate_in_channels  (rotate.py:66-73)
 *   `matmul(rotation.T, weight.view(heads, hc, -1).to(float64)).to(dtype)`        rotate_out_channels (rotate.py:76-95)
 *   `x.view(-1, k, n/k) @ hadamard_rhs`, `hadamard_lhs @ x`                       hardmard_transform  (hadamard.py:48-61)
 * ------------------------------------------------------------------------------------------- */
int dcq_gemm_f64(const void* A, int32_t a_storage, int64_t lda, int32_t a_kmajor, int64_t a_batch_stride, const void* B,
                 int32_t b_storage, int64_t ldb, int64_t b_batch_stride, void* C, int32_t c_storage, int64_t ldc,
                 int64_t c_batch_stride, int64_t M, int64_t N, int64_t K, int64_t batch, dcq_stream_t stream);

/* dcq_gram_inverse_factor_f64: X = R^-1 (upper triangular, zeros below) with R^T R = (G + G^T) / 2 for a float64
 * matrix G of order k <= 128, one workgroup with the matrix in LDS.  The small dense step of Cholesky QR in the
 * low-rank branch's eigen-solve, which keeps the `rank` leading triplets of the SVD the reference takes in full
 * (deepcompressor/nn/patch/lowrank.py:49-59); stands for `cholesky_ex(G, upper=True)` followed by
 * `solve_triangular(R, eye, upper=True)`.  *info: 0, or the 1-based index of the first non-positive pivot (X is
 * then meaningless).  X must not overlap G.  Larger orders: DCQ_ERR_UNSUPPORTED. */
int dcq_gram_inverse_factor_f64(const double* G, int64_t ldg, int32_t k, double* X, int64_t ldx, int32_t* info,
                                dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Smooth quantization (deepcompressor/calib/metric.py:31-86, calib/smooth.py:56-78, 852-891).
 *
 * dcq_channel_metric: x viewed as [N][C][inner] (float32 / float16 / bfloat16); per channel c, over all n and i:
 *   mode 0: max |x|          -> out_max[C] float32  (ChannelMetric._abs_max; caller zero-fills out_max; exact)
 *   mode 1: sum |x|          -> out_sum[C] float64  (ChannelMetric._abs_sum; caller zero-fills out_sum)
 *   mode 2: sum x^2          -> out_sum[C] float64  (ChannelMetric._square_sum)
 * Terms are formed in float32 as the reference does; workgroup partial sums (float32) are combined in float64.
 * Calls accumulate into the outputs, so a list of tensors is reduced by calling once per tensor.
 *
 * dcq_scale_channels: y[o][c][i] = storage(float32(x[o][c][i]) * scale[c])  (or / scale[c] when divide != 0) for
 * c < scaled_channels, a plain cast for the other channels: `param.to(float32).mul_(scale).to(dtype)`
 * (smooth_upscale_param), `narrow(dim, 0, scale.numel()).div_(scale)` (smooth_downscale_param) and
 * ActivationSmoother.process.  x and y may be the same buffer when the storage types match.
 * ------------------------------------------------------------------------------------------- */
/* dcq_smooth_fake_quant_groups: one smoothing candidate applied to a tensor in ONE pass (calib/smooth.py:504-530), or
 * the TensorError of K candidates from one read of the tensor (calib/search.py:776-790 with the candidate processing of
 * smooth.py:519-530).  x[G][S] holds groups that run along the channel dimension: element i of group g is channel
 * (g % groups_per_row) * S + i; chan_scale[K][C] with C = groups_per_row * S.
 *   mode 0 (weights):      y = storage(Q(float32(x) * s) / s)        `w.mul(s)` -> quantize -> `.div(s)` -> cast
 *   mode 1 (activations):  y = storage(Q(float32(x) / s) * s)        ActivationSmoother + quantizer + un-smoothing
 * Q is the single-step, single-scale-level integer fake quantization of `step` on the float32 values with a dynamic
 * per-group range (times `ratio` when has_ratio), exactly as dcq_fake_quant_groups computes it.
 *   dequant != NULL (K must be 1): y is written in the layout and storage type of x.
 *   errors  != NULL: errors[k][g] = sum_i pow(|y_k - x|, degree), difference, magnitude and power each rounded to the
 *                    storage type, the sum in float32 and left unrounded (as dcq_group_error); nothing else is written.
 * Exactly one of dequant / errors is given.  DCQ_ERR_UNSUPPORTED for step types the fused kernel does not cover. */
int dcq_smooth_fake_quant_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                                 int64_t groups_per_row, const float* chan_scale, int32_t K, int32_t mode,
                                 int32_t has_ratio, float ratio, void* dequant, float* errors, double degree,
                                 const uint16_t* pow_table, int32_t* nonfinite, dcq_stream_t stream);

/* dcq_smooth_quantize_groups: the same candidates (same arguments, same arithmetic up to the rounded code), but what
 * is written is what the GEMM below multiplies: codes[k][g][i] (int8) = the integer that the group's scale
 * multiplies -- the code, minus the zero point when the zero point lives in the code's domain -- and
 * scales[k][g] (float32) = the group's scale, for every candidate k.  The fake-quantized tensor of candidate k is
 * codes * scales, times (mode 1) or divided by (mode 0) chan_scale[k].  DCQ_ERR_UNSUPPORTED when the zero point is
 * applied after the scale, or code minus zero point can leave [-128, 127].
 *
 * dcq_linear_candidate_errors: K smoothing candidates of ONE linear module scored on its outputs (the OutputsError
 * objective when the evaluated module is the linear itself, calib/search.py:1011-1042 with smooth.py:575-625 putting
 * the candidate in place):
 *     errors[k] (+)= sum_{t, r} ( sum_c X'_k[t][c] W'_k[r][c] + bias[r] - target[t][r] )^2
 * with X'_k = Q(x / s_k) * s_k and W'_k = Q(w * s_k) / s_k given by their codes and group scales (x_codes [K][T][C],
 * x_scales [K][T][C / x_group], w_codes [K][R][C], w_scales [K][R][C / w_group]); s_k cancels in every product, so
 * the inner sums run over int8 codes on the matrix cores, exact in int32, and are scaled per (token group, row group)
 * in float32.  The reference multiplies the float32 processed tensors instead: both agree up to the float32
 * rounding of the processed values.  target [T][ld_target] holds the outputs to match (bias included); bias may be
 * NULL.  errors [K] is float64; accumulate != 0 adds to it (a list of batches is scored by calling once per batch).
 * C must be a multiple of 16; groups that are shorter than a row must be multiples of 128 channels with one size a
 * multiple of the other; at most 131072 channels per group of the finer side (the int32 sums).  workspace: dcq_linear_candidate_errors_workspace(T, R, K) bytes. */
int dcq_smooth_quantize_groups(const dcq_step* step, const void* x, int32_t storage, int64_t G, int64_t S,
                               int64_t groups_per_row, const float* chan_scale, int32_t K, int32_t mode,
                               int32_t has_ratio, float ratio, int8_t* codes, float* scales, int32_t* nonfinite,
                               dcq_stream_t stream);

int64_t dcq_linear_candidate_errors_workspace(int64_t T, int64_t R, int32_t K);

int dcq_linear_candidate_errors(const int8_t* x_codes, const float* x_scales, int64_t T, int64_t x_group,
                                const int8_t* w_codes, const float* w_scales, int64_t R, int64_t w_group, int64_t C,
                                int32_t K, const float* bias, const float* target, int64_t ld_target, void* workspace,
                                int64_t workspace_bytes, double* errors, int32_t accumulate, dcq_stream_t stream);

int dcq_channel_metric(const void* x, int32_t storage, int64_t N, int64_t C, int64_t inner, int32_t mode, float* out_max,
                       double* out_sum, dcq_stream_t stream);

int dcq_scale_channels(const void* x, int32_t storage, int64_t outer, int64_t C, int64_t inner, const float* scale,
                       int64_t scaled_channels, int32_t divide, void* y, int32_t out_storage, dcq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight packers for the serving back ends that read the checkpoints.
 *
 * dcq_pack_qserve_w4: QServePacker.pack_weight (deepcompressor/backend/qserve/utils.py:18-46): codes[n][k]
 * (uint8, values 0..15) -> out[n][k / 2] (int8) in the back end's fragment order, rows 16 apart sharing a byte
 * (high nibble: the second row).  n and k must be multiples of 32.
 * ------------------------------------------------------------------------------------------- */
int dcq_pack_qserve_w4(const uint8_t* codes, int64_t n, int64_t k, int8_t* out, dcq_stream_t stream);

/* dcq_pack_nunchaku: NunchakuWeightPacker.pack_weight (deepcompressor/backend/nunchaku/utils.py:16-62) for
 * bits = 4 or 8: codes[n][k] (int8, one code per byte; only the low `bits` bits are kept) -> out[n][k * bits / 8]
 * (int8) in the back end's fragment order.  n must be a multiple of 128, k of 128 (4 bits) or 64 (8 bits). */
int dcq_pack_nunchaku(const int8_t* codes, int64_t n, int64_t k, int32_t bits, int8_t* out, dcq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCQ_H_ */
