"""CPU restatement (numpy) of the reference's smooth quantization arithmetic.  TEST INFRASTRUCTURE ONLY: imported
by tests/, never by the product package.

Parity pinned: tests/golden/smooth.npz holds outputs of the reference's own functions
(deepcompressor/calib/metric.py:31-195, calib/smooth.py:81-146, 504-530, 852-1015) on seeded inputs, produced by
oracle/make_golden.py; tests/test_oracle_smooth.py checks this file against them.  Float32 sums and ``pow`` are
order / library dependent: those comparisons carry a stated tolerance, everything else is bit-exact.
"""

from __future__ import annotations

import numpy as np

import quant_oracle as qo

F32 = np.float32


def channel_metric(tensors, num_channels: int, mode: str) -> np.ndarray:
    """ChannelMetric.abs_max / abs_mean / root_mean_square over tensors viewed as [N, C, -1] (metric.py:142-195)."""
    tensors = [tensors] if isinstance(tensors, np.ndarray) else list(tensors)
    views = [t.reshape(t.shape[0], num_channels, -1).astype(F32) for t in tensors]
    if mode == "abs_max":
        return np.maximum.reduce([np.abs(v).max(axis=(0, 2)) for v in views]).astype(F32)
    count = sum(v.shape[0] * v.shape[2] for v in views)
    if mode == "abs_mean":
        total = sum(np.abs(v).astype(np.float64).sum(axis=(0, 2)) for v in views)
        return (total.astype(F32) / F32(count)).astype(F32)
    if mode == "root_mean_square":
        total = sum((v * v).astype(np.float64).sum(axis=(0, 2)) for v in views)
        return np.sqrt((total.astype(F32) / F32(count)).astype(F32)).astype(F32)
    raise ValueError(mode)


def smooth_scale(a_span: np.ndarray, b_span: np.ndarray, alpha: float, beta: float) -> np.ndarray:
    """get_smooth_scale (smooth.py:116-146): a^alpha / b^beta, zeros -> 1, any NaN / Inf -> all ones."""
    power = lambda base, e: np.power(base.astype(np.float64), e).astype(F32)  # noqa: E731
    if alpha > 0:
        s = power(a_span, alpha)
        if beta > 0:
            with np.errstate(divide="ignore", invalid="ignore"):
                s = (s / power(b_span, beta)).astype(F32)
    else:
        with np.errstate(divide="ignore"):
            s = power(b_span, -beta)
    s = np.where(s == 0, F32(1), s).astype(F32)
    if not np.isfinite(s).all():
        s = np.ones_like(s)
    return s


def _bcast(scale: np.ndarray, ndim: int, channels_dim: int) -> np.ndarray:
    shape = [1] * ndim
    shape[channels_dim] = -1
    return scale.astype(F32).reshape(shape)


def upscale(param: np.ndarray, dt: str, scale: np.ndarray, channels_dim: int = 1) -> np.ndarray:
    """smooth_upscale_param (smooth.py:852-869): float32 multiply, cast back."""
    return qo.round_to(dt, (param.astype(F32) * _bcast(scale, param.ndim, channels_dim)).astype(F32))


def downscale(param: np.ndarray, dt: str, scale: np.ndarray, channels_dim: int = 0) -> np.ndarray:
    """smooth_downscale_param (smooth.py:872-891): the first scale.size channels are divided."""
    out = param.astype(F32).copy()
    idx = [slice(None)] * param.ndim
    idx[channels_dim] = slice(0, scale.size)
    out[tuple(idx)] = (out[tuple(idx)] / _bcast(scale, param.ndim, channels_dim)).astype(F32)
    return qo.round_to(dt, out)


def downscale_for_producer(scale: np.ndarray, num_heads: int, num_head_repeats: int) -> np.ndarray:
    """convert_smooth_upscale_to_downscale (smooth.py:894-916)."""
    if num_heads > 1 and num_head_repeats > 1:
        hc = scale.size // num_heads
        return scale.reshape(num_heads // num_head_repeats, num_head_repeats, hc)[:, 0, :].reshape(-1)
    return scale


def candidate_error(weights, xs, scale, w_spec, x_spec, degree: float = 2) -> float:
    """Outputs error of one smoothing candidate for float32 linears sharing their input (smooth.py:504-530,
    search.py:817-830): sum over batches of (concat_i Q(x / s) s @ (Q(W_i s) / s)^T  -  concat_i x @ W_i^T)^degree."""
    total = 0.0
    s = scale.astype(F32)
    w_proc = []
    for w in weights:
        ws = (w.astype(F32) * s[None, :]).astype(F32)
        wq = qo.fake_quantize(ws, "f32", w_spec, default_dtype="f32")["dequant"]
        w_proc.append((wq / s[None, :]).astype(F32))
    for x in xs:
        xd = (x.astype(F32) / s).astype(F32)
        xq = qo.fake_quantize(xd.reshape(-1, xd.shape[-1]), "f32", x_spec, default_dtype="f32")["dequant"].reshape(x.shape)
        xp = (xq * s).astype(F32)
        y = np.concatenate([xp @ w.T for w in w_proc], axis=-1)
        y0 = np.concatenate([x.astype(F32) @ w.astype(F32).T for w in weights], axis=-1)
        total += float(np.sum(np.power((y - y0).astype(F32), degree), dtype=np.float64))
    return total


def _repeat_heads(span: np.ndarray, unique_heads: int, repeats: int) -> np.ndarray:
    return np.broadcast_to(span.reshape(unique_heads, 1, -1), (unique_heads, repeats, span.size // unique_heads)).reshape(-1)


def _share(span: np.ndarray, outer: int, inner: int, use_max: bool) -> np.ndarray:
    v = span.reshape(outer, inner, -1)
    r = v.max(axis=1, keepdims=True) if use_max else v.mean(axis=1, keepdims=True, dtype=F32)
    return np.broadcast_to(r, v.shape).reshape(-1).astype(F32)


def attention_spans(queries, keys, *, num_heads: int, num_head_repeats: int, with_rope: bool, key_mode: str,
                    query_mode: str) -> tuple[np.ndarray, np.ndarray]:
    """(alpha base, beta base) of the query / key smoothing search (smooth.py:300-383): key spans (one per key/value
    head, repeated over the query heads sharing it) and query spans (shared inside each group of query heads);
    with rotary embedding the two halves of every head share one span."""
    grouped = num_heads > 1 and num_head_repeats > 1
    unique = num_heads // num_head_repeats if grouped else 0
    mode = {"AbsMax": "abs_max", "RootMeanSquare": "root_mean_square"}
    q_span = channel_metric([q.reshape(-1, q.shape[-1]) for q in queries], queries[0].shape[-1], mode[query_mode])
    k_span = channel_metric([k.reshape(-1, k.shape[-1]) for k in keys], keys[0].shape[-1], mode[key_mode])
    if unique:
        q_span = _share(q_span, unique, num_head_repeats, "Max" in query_mode)
        k_span = _repeat_heads(k_span, unique, num_head_repeats)
    if with_rope:
        q_span = _share(q_span, num_heads, 2, "Max" in query_mode)
        k_span = _share(k_span, num_heads, 2, "Max" in key_mode)
    return k_span.astype(F32), q_span.astype(F32)


def attention_candidate_error(wq, wk, xs, scale, *, num_heads: int, num_head_repeats: int, head_dim: int, q_spec,
                              k_spec, degree: float = 2) -> float:
    """Outputs error of one query / key smoothing candidate on float32 score modules (smooth.py:627-653,
    search.py:1048-1077): q_proj rows * s, k_proj rows / s (first query head of each group), both products
    fake-quantized, error of the q k^T scores against the unscaled, unquantized ones."""
    s = scale.astype(F32)
    kv_heads = num_heads // num_head_repeats
    wq_s = (wq.astype(F32) * s[:, None]).astype(F32)
    wk_s = (wk.astype(F32) / downscale_for_producer(s, num_heads, num_head_repeats)[:, None]).astype(F32)

    def scores(x, a, b, quantize):
        bsz, t, _ = x.shape
        q, k = (x @ a.T).astype(F32), (x @ b.T).astype(F32)
        if quantize:
            if q_spec is not None:
                q = qo.fake_quantize(q.reshape(-1, q.shape[-1]), "f32", q_spec, default_dtype="f32")["dequant"].reshape(q.shape)
            if k_spec is not None:
                k = qo.fake_quantize(k.reshape(-1, k.shape[-1]), "f32", k_spec, default_dtype="f32")["dequant"].reshape(k.shape)
        q = q.reshape(bsz, t, num_heads, head_dim).transpose(0, 2, 1, 3)
        k = np.repeat(k.reshape(bsz, t, kv_heads, head_dim).transpose(0, 2, 1, 3), num_head_repeats, axis=1)
        return (q @ k.transpose(0, 1, 3, 2)).astype(F32)

    total = 0.0
    for x in xs:
        x = x.astype(F32)
        d = (scores(x, wq_s, wk_s, True) - scores(x, wq.astype(F32), wk.astype(F32), False)).astype(F32)
        total += float(np.sum(np.power(d, degree), dtype=np.float64))
    return total


def candidate_error_low_rank(weights, xs, scale, w_spec, x_spec, rank: int, *, exclusive: bool = False,
                             compensate: bool = False, splits=None, degree: float = 2) -> float:
    """Outputs error of one smoothing candidate scored with the low-rank branch in place (smooth.py:588-623):
    W s is split per group of modules into branch L and quantized remainder Q (quantize_with_low_rank); the module
    computes Q(x / s) Q^T + (x / s) L^T -- the branch's hook sits before the input quantizer's."""
    import lowrank_oracle as lo

    s = scale.astype(F32)
    scaled = [(w.astype(F32) * s[None, :]).astype(F32) for w in weights]
    bounds = [0, *(splits or []), len(weights)]
    parts = []
    for begin, end in zip(bounds[:-1], bounds[1:]):
        parts += lo.joint(scaled[begin:end], "f32", w_spec, rank, exclusive=exclusive, compensate=compensate)
    total = 0.0
    for x in xs:
        xd = (x.astype(F32) / s).astype(F32)
        xq = qo.fake_quantize(xd.reshape(-1, xd.shape[-1]), "f32", x_spec, default_dtype="f32")["dequant"].reshape(x.shape)
        y = np.concatenate([(xq @ q.T + xd @ l.T).astype(F32) for q, l in parts], axis=-1)
        y0 = np.concatenate([x.astype(F32) @ w.astype(F32).T for w in weights], axis=-1)
        total += float(np.sum(np.power((y - y0).astype(F32), degree), dtype=np.float64))
    return total


def products_errors(weights, xs, scale, w_spec, x_spec, dt: str, group: int, degree: float = 2):
    """ProductsError of one smoothing candidate per group of ``group`` input channels (search.py:791-816 with the
    reshapes of search.py:26-42): for every linear, sum over rows and tokens of
    (dt(x_g' @ w_g'^T) - dt(x_g @ w_g^T))^degree, where ' marks the processed tensors of smooth.py:504-530."""
    s = scale.astype(F32)
    out = []
    x_proc = []
    for x in xs:
        x2 = x.reshape(-1, x.shape[-1]).astype(F32)
        if x_spec is not None:
            xd = (x2 / s).astype(F32)
            xq = qo.fake_quantize(xd, "f32", x_spec, default_dtype=dt)["dequant"]
            x_proc.append(qo.round_to(dt, (xq * s).astype(F32)))
        else:
            x_proc.append(x2)
    for w in weights:
        ws = (w.astype(F32) * s[None, :]).astype(F32)
        wq = qo.fake_quantize(ws, "f32", w_spec, default_dtype=dt)["dequant"]
        wp = qo.round_to(dt, (wq / s[None, :]).astype(F32))
        rows, cols = w.shape
        total = np.zeros(cols // group, np.float64)
        for x, xp in zip(xs, x_proc):
            x2 = x.reshape(-1, cols).astype(F32)
            part = lambda a, b: qo.round_to(dt, np.einsum(  # noqa: E731
                "ngi,rgi->rgn", a.reshape(-1, cols // group, group), b.reshape(rows, cols // group, group)).astype(F32))
            diff = qo.round_to(dt, part(xp, wp) - part(x2, w.astype(F32)))
            total += np.power(diff.astype(np.float64), degree).sum(axis=(0, 2))
        out.append(total)
    return out


def tensor_errors(weights, scale, w_spec, dt: str, group: int, degree: float = 2, granularity: str = "ChannelGroup"):
    """TensorError of one smoothing candidate (search.py:776-790 with the weight processing of smooth.py:504-530): for
    every linear, |dt(Q(w s) / s) - w|^degree evaluated in the weights' dtype and summed per group of ``group`` input
    channels over all rows (ChannelGroup) or over the whole weight (Layer)."""
    s = scale.astype(F32).reshape(-1)
    out = []
    for w in weights:
        ws = (w.astype(F32) * s[None, :]).astype(F32)
        wq = qo.fake_quantize(ws, "f32", w_spec, default_dtype=dt)["dequant"]
        wp = qo.round_to(dt, (wq / s[None, :]).astype(F32))
        e = np.abs(qo.round_to(dt, wp - w.astype(F32)))
        e = qo.pow_like_torch(e, degree, dt)
        rows, cols = w.shape
        out.append(qo._granular_error(e, (rows, 1, cols // group, group), granularity, dt))
    return out


def update_best(best_error, best_scale, error, scale, num_channels: int, num_heads: int = 1, num_head_repeats: int = 1):
    """SmoothCalibrator._update_best (smooth.py:655-714) on lists of per-group error vectors."""
    if best_error is None:
        return [e.copy() for e in error], scale.copy()
    numel = error[0].size
    if numel == 1:
        if all(float(e.reshape(-1)[0]) <= float(b.reshape(-1)[0]) for b, e in zip(best_error, error)):
            return [e.copy() for e in error], scale.copy()
        return best_error, best_scale
    group = num_channels // numel
    shared = num_heads > 1 and num_head_repeats > 1
    if shared:
        head_channels, unique_heads = num_channels // num_heads, num_heads // num_head_repeats
        per_head, repeats, unique_per_group = 1, 1, 1
        if head_channels >= group:
            per_head, repeats = head_channels // group, num_head_repeats
        else:
            heads_per_group = group // head_channels
            if heads_per_group < num_head_repeats:
                repeats = num_head_repeats // heads_per_group
            else:
                unique_per_group = heads_per_group // num_head_repeats
        uniques = unique_heads // unique_per_group
        shared = repeats > 1
    pos = np.ones(numel, bool)
    for b, e in zip(best_error, error):
        if shared:
            bs = b.reshape(uniques, repeats, per_head).sum(axis=1, keepdims=True, dtype=b.dtype)
            es = e.reshape(uniques, repeats, per_head).sum(axis=1, keepdims=True, dtype=e.dtype)
            pos &= np.broadcast_to(es < bs, (uniques, repeats, per_head)).reshape(-1)
        else:
            pos &= e < b
    for b, e in zip(best_error, error):
        b[pos] = e[pos]
    best_scale = best_scale.copy().reshape(numel, group)  # the reference hands back this 2-D view (smooth.py:712-714)
    best_scale[pos] = scale.reshape(numel, group)[pos]
    return best_error, best_scale
