"""CPU restatement (numpy float32 + LAPACK) of the reference's GPTQ kernel.

TEST INFRASTRUCTURE ONLY (see quant_oracle.py).  Follows deepcompressor/quantizer/kernel/gptq.py:128-285 step
by step.  GPTQ results are not bit-reproducible across BLAS/LAPACK builds (the Hessian GEMM and the Cholesky
factorisations sum in library-specific orders), so parity for this path is tolerance based: tests compare
codes (fraction identical) and the layer output x @ W_q^T (relative error), against the fixtures captured from
the reference in tests/golden/gptq.npz.
"""

from __future__ import annotations

import math

import numpy as np
import scipy.linalg

import quant_oracle as qo

F32 = np.float32


def hessian(x: np.ndarray, num_samples: int, block: int = 512) -> np.ndarray:
    """H = sum_b (sqrt(2/n) x_b)^T (sqrt(2/n) x_b) in float32 (gptq.py:184-194)."""
    C = x.shape[1]
    H = np.zeros((C, C), dtype=F32)
    s = F32(math.sqrt(2 / num_samples))
    step = block if block > 0 else x.shape[0]
    for b in range(0, x.shape[0], step):
        xb = (x[b : b + step].astype(F32) * s).astype(F32)
        H += xb.T @ xb
    return H


def inverse_factor(H: np.ndarray, damp: float, tries: int = 200):
    """Damping + cholesky -> cholesky_inverse -> upper cholesky with the retry rule (gptq.py:210-233).  H is modified."""
    d = np.einsum("ii->i", H)
    mean = d.mean(dtype=F32)
    d += F32(damp) * mean
    for _ in range(tries):
        try:
            L = np.linalg.cholesky(H)
            inv, info = scipy.linalg.lapack.spotri(L, lower=1)
            assert info == 0
            inv = np.tril(inv) + np.tril(inv, -1).T
            return np.ascontiguousarray(np.linalg.cholesky(inv.astype(F32)).T)
        except (np.linalg.LinAlgError, AssertionError):
            d += F32(damp * 0.1) * mean
    raise RuntimeError("hessian not invertible")


def gptq_codes(w: np.ndarray, x: np.ndarray, num_samples: int, step: qo.StepSpec, scale: np.ndarray, zero: np.ndarray,
               view_shape, *, damp=0.01, block=128, hessian_block=512, quant_range=None, range_bound=None,
               H: np.ndarray | None = None) -> np.ndarray:
    """Codes (float32, original column order, shape (R, C)) for the float32 working weight ``w`` (R, C)."""
    n_rg, rg, n_cg, cg = view_shape
    R, C = n_rg * rg, n_cg * cg
    W = np.array(w, dtype=F32).reshape(R, C)
    H = hessian(x, num_samples, hessian_block) if H is None else np.array(H, dtype=F32)
    dead = np.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    perm = np.argsort(-np.diag(H), kind="stable")
    H = np.ascontiguousarray(H[perm][:, perm])
    W = np.ascontiguousarray(W[:, perm])
    inv_perm = np.argsort(perm)
    U = inverse_factor(H, damp)
    s2 = np.broadcast_to(np.asarray(scale, dtype=F32).reshape(n_rg, 1, n_cg, 1) if np.size(scale) > 1 else scale,
                         (n_rg, 1, n_cg, 1)).reshape(n_rg, n_cg)
    zero = np.asarray(zero, dtype=F32)
    z2 = None if zero.size == 1 else zero.reshape(n_rg, n_cg)
    Q = np.zeros_like(W)
    zd = step.zero_domain
    for c0 in range(0, C, block):
        c1 = min(c0 + block, C)
        blk = W[:, c0:c1].copy()
        err = np.zeros_like(blk)
        Ub = U[c0:c1, c0:c1]
        for j in range(c1 - c0):
            col = blk[:, j]
            g = perm[c0 + j] // cg
            s = np.repeat(s2[:, g], rg)
            z = np.repeat(z2[:, g], rg) if z2 is not None else zero.reshape(())
            t = col.copy()
            if range_bound is not None:
                t = np.clip(t, None if range_bound[0] is None else F32(range_bound[0]),
                            None if range_bound[1] is None else F32(range_bound[1]))
            if zd == "PostScale":
                t = (t + z).astype(F32)
            t = (t / s).astype(F32)
            if zd == "PreScale":
                t = (t + z).astype(F32)
            q = qo.simple_quantize(t, step.dtype, has_zero_point=zd is not None, qr=quant_range)
            Q[:, c0 + j] = q
            d = q
            if zd == "PreScale":
                d = (d - z).astype(F32)
            d = (d * s).astype(F32)
            if zd == "PostScale":
                d = (d - z).astype(F32)
            e = ((col - d) / Ub[j, j]).astype(F32)
            err[:, j] = e
            blk[:, j:] -= np.outer(e, Ub[j, j:]).astype(F32)
        if c1 < C:
            W[:, c1:] -= err @ U[c0:c1, c1:]
    return Q[:, inv_perm]
