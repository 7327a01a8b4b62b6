"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of deepcompressor's channel reordering
(deepcompressor/calib/reorder.py:61-540, calib/metric.py:13-195).  Imported by tests only; the product never
imports this module.  Pinned by ``tests/golden/reorder.npz`` (generated from the reference by
``oracle/make_golden.py``)."""

from __future__ import annotations

import numpy as np

import quant_oracle as qo

F32 = np.float32
METRICS = ("InputsAbsMax", "InputsAbsMean", "InputsRootMeanSquare", "WeightsAbsMax", "WeightsAbsMean",
           "WeightsRootMeanSquare", "AbsMaxProduct", "AbsMeanProduct", "RootMeanSquareProduct")
INDEX_MODES = ("Sequential", "Transpose")


def _stat(tensors: list[np.ndarray], kind: str, channels: int) -> np.ndarray:
    """calib/metric.py:31-195 on tensors viewed as [N, channels, -1], float32 like the reference's develop dtype."""
    views = [t.reshape(t.shape[0], channels, -1).astype(F32) for t in tensors]
    if kind == "AbsMax":
        return np.max([np.abs(v).max(axis=(0, 2)) for v in views], axis=0).astype(F32)
    count = sum(v.shape[0] * v.shape[2] for v in views)
    if kind == "AbsMean":
        total = sum(np.abs(v).astype(np.float64).sum(axis=(0, 2)) for v in views)
        return (total.astype(F32) / F32(count)).astype(F32)
    assert kind == "RootMeanSquare"
    total = sum((v.astype(np.float64) ** 2).sum(axis=(0, 2)) for v in views)
    return np.sqrt((total.astype(F32) / F32(count)).astype(F32)).astype(F32)


def abs_normalize_mean(tensors: list[np.ndarray], channels: int, group_channels: int) -> np.ndarray:
    """calib/metric.py:16-30, 169-181 for 2-D tensors and groups (-1, group_channels)."""
    out = []
    for t in tensors:
        v = t.astype(F32).reshape(1, t.shape[0], channels // group_channels, group_channels)
        m = np.abs(v).max(axis=(1, 3), keepdims=True)
        m[m == 0] = 1
        out.append((v / m).astype(F32).reshape(t.shape))
    return _stat(out, "AbsMean", channels)


def channel_metric(xs: list[np.ndarray], ws: list[np.ndarray], mode: str, channels: int) -> np.ndarray:
    """reorder.py:91-157; activations flattened to [tokens, channels]."""
    if mode.endswith("Product"):
        kind = mode[:-7]
        return (channel_metric(xs, ws, f"Inputs{kind}", channels) * channel_metric(xs, ws, f"Weights{kind}", channels)).astype(F32)
    if mode.startswith("Inputs"):
        return _stat([x.reshape(-1, channels) for x in xs], mode[6:], channels)
    return _stat(list(ws), mode[7:], channels)


def index_from_rank(rank: np.ndarray, num_channels: int, num_groups: int, index_mode: str) -> np.ndarray:
    """reorder.py:61-88."""
    if index_mode == "Transpose":
        return rank.reshape(num_channels // num_groups, num_groups).T.reshape(-1)
    return rank


def index_from_metric(metric: np.ndarray, metric_mode: str, index_mode: str, group_size: int, heads: int = 1,
                      repeats: int = 1) -> np.ndarray:
    """reorder.py:211-265 (ascending, stable)."""
    n = metric.size
    groups = n // group_size
    if heads > 1:
        hc = n // heads
        if repeats > 1:
            m = metric.reshape(heads // repeats, repeats, hc)
            m = m.max(axis=1, keepdims=True) if "Max" in metric_mode else m.sum(axis=1, keepdims=True, dtype=F32)
            rank = np.broadcast_to(np.argsort(m, axis=-1, kind="stable"), (heads // repeats, repeats, hc)).reshape(heads, hc)
        else:
            rank = np.argsort(metric.reshape(heads, hc), axis=-1, kind="stable")
        rank = rank + np.arange(0, n, hc).reshape(heads, 1)
        return np.concatenate([index_from_rank(rank[h], hc, max(groups // heads, 1), index_mode) for h in range(heads)])
    return index_from_rank(np.argsort(metric, kind="stable"), n, groups, index_mode)


def candidates(xs, ws, channels: int, group_size: int, heads: int = 1, repeats: int = 1, manual=None):
    """The calibrator's candidate list (reorder.py:357-400): [identity] + metric-major, index-mode-minor for a search;
    the one configured (metric, index mode) for Manual."""
    if manual is not None:
        mm, im = manual
        return [index_from_metric(_heads_metric(xs, ws, mm, channels), mm, im, group_size, heads, repeats)]
    out = [None]
    for mm in METRICS:
        metric = _heads_metric(xs, ws, mm, channels)
        for im in INDEX_MODES:
            out.append(index_from_metric(metric, mm, im, group_size, heads, repeats))
    return out


def _heads_metric(xs, ws, mode, channels):
    return channel_metric(xs, ws, mode, channels)  # the head count only shapes the (unused) normalisation groups


def search(cands, evaluate, degree: float = 2):
    """reorder.py:413-449 with the OutputsError scoring of calib/search.py:817-832.  ``evaluate(index)`` returns
    (outputs per batch with the candidate order in place and everything fake-quantized, baseline outputs per batch).
    Returns (best candidate id, root errors per candidate)."""
    roots, base_root, best, best_id = [], None, None, None
    for cid, idx in enumerate(cands):
        outs, base = evaluate(idx)
        total = F32(0)
        for y, y0 in zip(outs, base):
            d = (y - y0).astype(F32)
            pw = (d * d) if degree == 2 else np.power(d, F32(degree)).astype(F32)
            total = F32(total + F32(pw.astype(np.float64).sum()))
        root = float(np.float64(total) ** (1.0 / degree))
        roots.append(root)
        if base_root is None:
            base_root = root
        d = root - base_root
        stats = [int(root > base_root), -int(root < base_root), max(d, 0), min(d, 0), root]
        if best is None or stats < best:
            best, best_id = stats, cid
    return best_id, roots


def quantize_reordered(t: np.ndarray, index, spec, dtype: str) -> np.ndarray:
    """index_select -> fake-quantize -> inverse index_select along the last axis (reorder.py:472-492)."""
    t2 = t.reshape(-1, t.shape[-1])
    if index is not None:
        t2 = t2[:, index]
    q = qo.fake_quantize(np.ascontiguousarray(t2), dtype, spec, default_dtype=dtype)["dequant"]
    if index is not None:
        inv = np.empty_like(index)
        inv[index] = np.arange(index.size)
        q = q[:, inv]
    return q.reshape(t.shape)
