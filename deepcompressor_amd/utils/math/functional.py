"""Small math helpers (reference: deepcompressor/utils/math/functional.py:6-39)."""

import torch

__all__ = ["is_pow2", "root_"]


def is_pow2(n: int) -> bool:
    """True for 1, 2, 4, 8, ..."""
    return n > 0 and n & (n - 1) == 0


def root_(y: torch.Tensor, index: float) -> torch.Tensor:
    """In-place ``index``-th root; the square root takes ``sqrt_`` (correctly rounded), others ``pow_``."""
    return y.sqrt_() if index == 2 else y.pow_(1 / index)
