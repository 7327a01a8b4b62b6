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
