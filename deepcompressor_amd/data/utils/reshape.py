"""Reshape callables that turn module inputs into 2-D GEMM operands
(reference: deepcompressor/data/utils/reshape.py:16-160; the convolution variants are not part of the
linear-layer calibration path and are not provided)."""

import torch

__all__ = ["ReshapeFn", "LinearReshapeFn"]


class ReshapeFn:
    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        return x


class LinearReshapeFn(ReshapeFn):
    """(..., C) -> (N, C), or (C, N) when ``ic_last`` is false."""

    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        flat = x.view(-1, x.shape[-1])
        return flat if ic_last else flat.t()
