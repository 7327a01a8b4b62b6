"""CPU restatement (numpy) of the reference's weight rotations.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product package.

Parity pinned: tests/golden/rotate.npz holds outputs of the reference's own functions
(deepcompressor/calib/rotate.py:66-133, utils/math/hadamard.py:17-103) on seeded inputs, produced by
oracle/make_golden.py; tests/test_oracle_rotate.py checks this file against them.
"""

from __future__ import annotations

import numpy as np

import quant_oracle as qo


def cast_from_f64(x64: np.ndarray, dt: str) -> np.ndarray:
    """Tensor.to(dtype) of a float64 tensor: double -> float (round to nearest even), then float -> 16 bit."""
    return qo.round_to(dt, x64.astype(np.float32))


def rotate_in_channels(w: np.ndarray, dt: str, rotation: np.ndarray) -> np.ndarray:
    """rotate.py:66-73: (w.view(-1, n).double() @ rotation).to(dtype)."""
    n = rotation.shape[0]
    return cast_from_f64(w.reshape(-1, n).astype(np.float64) @ rotation, dt).reshape(w.shape)


def rotate_out_channels(w: np.ndarray, dt: str, rotation: np.ndarray) -> np.ndarray:
    """rotate.py:76-95: per head, rotation^T @ w_head (also used for the bias viewed as [heads, hc, 1])."""
    hc = rotation.shape[0]
    heads = w.shape[0] // hc
    out = np.matmul(rotation.T[None], w.reshape(heads, hc, -1).astype(np.float64))
    return cast_from_f64(out, dt).reshape(w.shape)


def sylvester(n: int) -> np.ndarray:
    h = np.ones((1, 1))
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h


def hadamard_transform(x64: np.ndarray, rhs: np.ndarray, lhs: np.ndarray | None, k: int, scaled: bool) -> np.ndarray:
    """hadamard.py:48-61 on float64 data."""
    n = x64.shape[-1]
    y = x64.reshape(-1, k, n // k) @ rhs
    if not scaled:
        y = y * (1.0 / np.sqrt(np.float64(n)))
    if k > 1:
        y = lhs @ y
    return y.reshape(x64.shape)


def butterflies(x: np.ndarray, hadamard_k: np.ndarray | None, k: int) -> np.ndarray:
    """hadamard.py:17-35 (`_matmul_hadU`): pairwise sum / difference passes down to blocks of k, the order-k
    matrix, then division by the float32 square root of n."""
    n = x.shape[-1]
    cur = x.reshape(-1, n, 1).astype(np.float64)
    while cur.shape[1] > k:
        pairs = cur.reshape(cur.shape[0], cur.shape[1] // 2, 2, cur.shape[2])
        cur = np.concatenate((pairs[:, :, 0] + pairs[:, :, 1], pairs[:, :, 0] - pairs[:, :, 1]), axis=-1)
    if k > 1:
        cur = hadamard_k.astype(np.float64)[None] @ cur
    return cur.reshape(x.shape) / np.float64(np.sqrt(np.float32(n)))
