"""CPU restatement (numpy) of the reference's low-rank branch arithmetic.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by the product package.

Parity pinned: tests/golden/lowrank.npz holds outputs of the reference's own code
(deepcompressor/nn/patch/lowrank.py:29-67, quantizer/processor.py:215-310, calib/lowrank.py:80-235) on seeded
inputs, produced by oracle/make_golden.py; tests/test_oracle_lowrank.py checks this file against them.  The SVD
comes from LAPACK (numpy here, torch there): singular vectors agree up to sign, so comparisons are made on the
product b @ a with a float tolerance.
"""

from __future__ import annotations

import numpy as np

import quant_oracle as qo

F32 = np.float32


def branch(w: np.ndarray, dt: str, rank: int):
    """LowRankBranch.reset_parameters (lowrank.py:29-59): float64 SVD, a = Vh[:r], b = U[:, :r] S[:r], cast to dt.
    Returns (a, b, b @ a) with the product formed in dt arithmetic (float32 accumulate, rounded to dt)."""
    u, s, vh = np.linalg.svd(w.astype(np.float64), full_matrices=False)
    a = qo.round_to(dt, vh[:rank].astype(F32))
    b = qo.round_to(dt, (u[:, :rank] * s[:rank]).astype(F32))
    return a, b, qo.round_to(dt, (b.astype(F32) @ a.astype(F32)).astype(F32))


def joint(weights, dt: str, spec, rank: int, *, exclusive: bool, compensate: bool):
    """Quantizer.quantize_with_low_rank (processor.py:215-310) -> list of (dequantized weight, branch weight)."""
    fq = lambda x: qo.fake_quantize(x, dt, spec, default_dtype=dt)["dequant"]  # noqa: E731
    out = []
    if len(weights) == 1 or exclusive:
        for w in weights:
            if compensate:
                q = fq(w)
                out.append((q, branch(qo.round_to(dt, w - q), dt, rank)[2]))
            else:
                low = branch(w, dt, rank)[2]
                out.append((fq(qo.round_to(dt, w - low)), low))
        return out
    stacked = np.concatenate(weights, axis=0)
    qs = [fq(w) for w in weights] if compensate else None
    if compensate:
        stacked = qo.round_to(dt, stacked - np.concatenate(qs, axis=0))
    a, b, _ = branch(stacked, dt, rank)
    start = 0
    for i, w in enumerate(weights):
        low = qo.round_to(dt, (b[start:start + w.shape[0]].astype(F32) @ a.astype(F32)).astype(F32))
        start += w.shape[0]
        out.append((qs[i], low) if compensate else (fq(qo.round_to(dt, w - low)), low))
    return out


def calibrate_errors(w: np.ndarray, xs, w_spec, x_spec, rank: int, num_iters: int, degree: float = 2):
    """QuantLowRankCalibrator on one float32 linear, OutputsError (lowrank.py:118-154, search.py:817-830):
    per iteration L = SVD_r(W - Q), Q = fq(W - L), error = sum ((fq(x) Q^T + x L^T) - x W^T)^degree: the branch's
    hook is registered before the input quantizer's, so the branch sees the unquantized input."""
    fq_w = lambda t: qo.fake_quantize(t, "f32", w_spec, default_dtype="f32")["dequant"]  # noqa: E731
    fq_x = lambda t: qo.fake_quantize(t.reshape(-1, t.shape[-1]), "f32", x_spec, default_dtype="f32")["dequant"].reshape(t.shape)  # noqa: E731
    qw = np.zeros_like(w)
    errors, lows = [], []
    for _ in range(num_iters):
        a, b, low = branch((w - qw).astype(F32), "f32", rank)
        qw = fq_w((w - low).astype(F32))
        total = 0.0
        for x in xs:
            xq = fq_x(x.astype(F32))
            y = xq @ qw.T + (x.astype(F32) @ a.T) @ b.T
            total += float(np.sum(np.power((y - x @ w.T).astype(F32), degree), dtype=np.float64))
        errors.append(total)
        lows.append(low)
    return np.array(errors), lows
