"""Scale-level helpers (reference: deepcompressor/data/utils/scale.py:13-55)."""

from __future__ import annotations

import typing as tp

import torch

from ..dtype import QuantDataType

__all__ = ["infer_scale_dtypes", "infer_scale_quant_spans", "infer_exponent_scale_level"]


def infer_scale_dtypes(scale_dtypes, default_dtype):
    assert isinstance(default_dtype, (torch.dtype, QuantDataType)), f"bad default dtype {default_dtype}"
    return [d or default_dtype for d in scale_dtypes]


def infer_scale_quant_spans(scale_dtypes: tp.Sequence[QuantDataType], base: int = 1) -> list[float]:
    """Span each level may use: the product of the max values of all finer (quantized) levels."""
    spans: list[float] = [base]
    for d in reversed(scale_dtypes[1:]):
        assert isinstance(d, QuantDataType), f"s_dtype must be QuantDataType, got {d}"
        spans.append(d.max_value * spans[-1])
    return spans[::-1]


def infer_exponent_scale_level(scale_dtypes) -> int:
    for level, d in enumerate(scale_dtypes):
        if isinstance(d, QuantDataType) and d.is_exponent:
            return level
    return len(scale_dtypes)
