"""Hadamard matrices and transforms (reference: deepcompressor/utils/math/hadamard.py:17-103).

Same call surface as the reference.  Two differences in how the work is done:

* the reference stores the non-power-of-two matrices (orders 12 ... 172) as literal tables; here they are
  *constructed*: Paley I for 12, 20, 60, 108, 140 (q = order - 1 prime, q = 3 mod 4), Paley II for 28 and 36
  (q = order / 2 - 1 prime, q = 1 mod 4), Sylvester doubling of the order-20 matrix for 40, and Williamson arrays
  of four symmetric circulants for 52, 156, 172 (their first rows below are the only stored data).  The test
  suite checks every constructed matrix entry by entry against the reference's tables;
* on a GPU ``hardmard_transform`` runs its two products through ``dcq_gemm_f64`` (float64 accumulation on the
  matrix cores, one rounding to the tensor's dtype per product).
"""

from __future__ import annotations

import ctypes as C
import math
import typing as tp

import numpy as np
import torch

from .functional import is_pow2

__all__ = ["random_hadamard_matrix", "HadamardMatrix", "hardmard_transform", "gemm_f64"]

_WILLIAMSON_ROWS = {  # first rows of the four symmetric circulant blocks A, B, C, D
    52: ("+-+--++++--+-", "+---++++++---", "++-+--++--+-+", "+----+--+----"),
    156: ("+++--+-+-----+--++----++--+-----+-+--++", "++++---+--++----+-+--+-+----++--+---+++",
          "+++--++-+---+-+--+----+--+-+---+-++--++", "+---++-+-+-----+++-++-+++-----+-+-++---"),
    172: ("+---++--++++-+-+++-++--++-+++-+-++++--++---", "++-++++++----+-+--++-++-++--+-+----++++++-+",
          "+++-+-++--+-+-++++-+----+-++++-+-+--++-+-++", "++---++++-+--+--++--------++--+--+-++++---+"),
}


def _sylvester(n: int) -> np.ndarray:
    h = np.ones((1, 1), dtype=np.int8)
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h


def _quadratic_character(q: int) -> np.ndarray:
    chi = -np.ones(q, dtype=np.int8)
    chi[0] = 0
    chi[(np.arange(1, q, dtype=np.int64) ** 2) % q] = 1
    return chi


def _paley1(q: int) -> np.ndarray:
    """Order q + 1: a column of +1, a first row of -1 (after the corner) and the circulant core I - Q."""
    chi = _quadratic_character(q)
    idx = (np.arange(q)[None, :] - np.arange(q)[:, None]) % q  # j - i
    core = -chi[idx]
    core[np.arange(q), np.arange(q)] = 1
    h = np.ones((q + 1, q + 1), dtype=np.int8)
    h[0, 1:] = -1
    h[1:, 1:] = core
    return h


def _paley2(q: int) -> np.ndarray:
    """Order 2 (q + 1) from the symmetric conference matrix S: [[S + I, S - I], [S - I, -S - I]]."""
    chi = _quadratic_character(q)
    s = np.zeros((q + 1, q + 1), dtype=np.int8)
    s[0, 1:] = 1
    s[1:, 0] = 1
    s[1:, 1:] = chi[(np.arange(q)[:, None] - np.arange(q)[None, :]) % q]
    eye = np.eye(q + 1, dtype=np.int8)
    return np.block([[s + eye, s - eye], [s - eye, -s - eye]])


def _williamson(rows: tp.Sequence[str]) -> np.ndarray:
    m = len(rows[0])
    idx = (np.arange(m)[None, :] - np.arange(m)[:, None]) % m
    a, b, c, d = (np.array([1 if ch == "+" else -1 for ch in row], dtype=np.int8)[idx] for row in rows)
    return np.block([[a, b, c, d], [-b, a, -d, c], [-c, d, a, -b], [-d, -c, b, a]])


def _hadamard_k(k: int) -> np.ndarray:
    if is_pow2(k):
        return _sylvester(k)
    if k in (12, 20, 60, 108, 140):
        return _paley1(k - 1)
    if k in (28, 36):
        return _paley2(k // 2 - 1)
    if k == 40:
        return np.kron(_sylvester(2), _paley1(19))
    if k in _WILLIAMSON_ROWS:
        return _williamson(_WILLIAMSON_ROWS[k])
    raise ValueError(f"no Hadamard matrix of order {k} is built in")


def gemm_f64(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype, *, a_transposed: bool = False) -> torch.Tensor:
    """``(a @ b).to(out_dtype)`` with float64 accumulation on the GPU: ``a`` is [batch?, M, K] (or [K, M] holding
    a^T when ``a_transposed``), ``b`` is [batch?, K, N]; a 2-D operand is shared by the whole batch.  Inputs of any
    float dtype are widened exactly; the result is rounded double -> float -> out_dtype."""
    from ..._lib import check, lib, stream_of

    storage = {torch.float64: 3, torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    if not (a.is_cuda and b.is_cuda):
        from ..._lib import DcqError
        raise DcqError("deepcompressor_amd runs on MI355X GPUs only (there is no CPU path)")
    a, b = a.contiguous(), b.contiguous()
    batch = a.shape[0] if a.ndim == 3 else (b.shape[0] if b.ndim == 3 else 1)
    a2, b2 = a.shape[-2:], b.shape[-2:]
    m, k = (a2[1], a2[0]) if a_transposed else (a2[0], a2[1])
    assert b2[0] == k, f"inner dimensions differ: {tuple(a.shape)} x {tuple(b.shape)}"
    n = b2[1]
    out_shape = (batch, m, n) if (a.ndim == 3 or b.ndim == 3) else (m, n)
    out = torch.empty(out_shape, dtype=out_dtype, device=b.device)
    if out.numel() == 0:
        return out
    raw = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    check(lib().dcq_gemm_f64(raw(a), storage[a.dtype], a2[1], int(a_transposed), a2[0] * a2[1] if a.ndim == 3 else 0,
                             raw(b), storage[b.dtype], n, k * n if b.ndim == 3 else 0, raw(out), storage[out_dtype], n,
                             m * n, m, n, k, batch, stream_of(b.device)), "dcq_gemm_f64")
    return out


def _matmul_hadU(X: torch.Tensor, hadamard_K: torch.Tensor | None, K: int) -> torch.Tensor:
    """Fast Walsh-Hadamard butterflies down to blocks of K, the order-K matrix on what remains, then the
    1 / sqrt(n) scaling (hadamard.py:17-35).  Entries stay small integers until the final division, so the result
    does not depend on where or in which order the additions run."""
    n = X.shape[-1]
    cur = X.reshape(-1, n, 1).clone()
    while cur.shape[1] > K:
        pairs = cur.view(cur.shape[0], cur.shape[1] // 2, 2, cur.shape[2])
        cur = torch.cat((pairs[:, :, 0] + pairs[:, :, 1], pairs[:, :, 0] - pairs[:, :, 1]), dim=-1)
    if K > 1:
        assert hadamard_K is not None
        cur = hadamard_K.view(1, K, K).to(cur) @ cur
    return cur.reshape(X.shape) / torch.tensor(n).sqrt()  # float32 square root, as in the reference


def random_hadamard_matrix(size: int) -> torch.Tensor:
    """A Hadamard matrix with random row signs, float64 (hadamard.py:38-45; signs from torch's global generator)."""
    signs = torch.randint(low=0, high=2, size=(size,)).to(torch.float64) * 2 - 1
    hadamard_K, K = HadamardMatrix.get_lhs(size)
    return _matmul_hadU(torch.diag(signs), hadamard_K=hadamard_K, K=K)


def hardmard_transform(x: torch.Tensor, hadamard_rhs: torch.Tensor, hadamard_lhs: torch.Tensor | None, lhs_k: int,
                       scaled: bool = False) -> torch.Tensor:
    """``x`` [.., n] viewed as [.., lhs_k, n / lhs_k]: right-multiply by ``hadamard_rhs``, scale by 1 / sqrt(n)
    unless the matrices are pre-scaled, left-multiply by ``hadamard_lhs`` (hadamard.py:48-61)."""
    shape = x.shape
    n = shape[-1]
    rhs_k = n // lhs_k
    y = gemm_f64(x.reshape(-1, rhs_k), hadamard_rhs.to(x.device), x.dtype)
    if not scaled:
        y = y.mul_((1.0 / torch.tensor(n, dtype=torch.float64).sqrt()).to(device=y.device, dtype=y.dtype))
    if lhs_k > 1:
        y = gemm_f64(hadamard_lhs.to(x.device), y.view(-1, lhs_k, rhs_k), x.dtype)
    return y.view(shape)


class HadamardMatrix:
    _cache: tp.ClassVar[dict] = {}

    @staticmethod
    def get(n: int, *, scale: bool, dtype: torch.dtype = torch.float32,
            device: torch.device | str = "cpu") -> tuple[torch.Tensor, torch.Tensor, int]:
        """(rhs, lhs, k) with n = k * 2^m: rhs the Sylvester matrix of order n / k (times 1 / sqrt(n) when
        ``scale``), lhs the order-k matrix (hadamard.py:71-86)."""
        device = torch.device(device)
        key = (n, scale, dtype, device)
        if key not in HadamardMatrix._cache:
            lhs, k = HadamardMatrix.get_lhs(n)
            rhs = torch.from_numpy(_sylvester(n // k).astype(np.float32))
            if scale:
                rhs = rhs.to(dtype=torch.float64).mul_(1.0 / torch.tensor(n, dtype=torch.float64).sqrt_())
            HadamardMatrix._cache[key] = (rhs.to(dtype=dtype, device=device), lhs.to(dtype=dtype, device=device), k)
        return HadamardMatrix._cache[key]

    @staticmethod
    def get_lhs(n: int) -> tuple[torch.Tensor, int]:
        """The largest built-in odd-multiple order that divides n (same preference order as hadamard.py:88-96)."""
        for k in (172, 156, 140, 108, 60, 52, 36, 28, 40, 20, 12):
            if n % k == 0:
                assert is_pow2(n // k)
                return HadamardMatrix._get_hadamard_k(k), k
        assert is_pow2(n)
        return torch.FloatTensor([1]), 1

    @staticmethod
    def _get_hadamard_k(k: int) -> torch.Tensor:
        return torch.from_numpy(_hadamard_k(k).astype(np.float32))
