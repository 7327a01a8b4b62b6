"""Element-wise snapping of a tensor onto a data type
(reference: deepcompressor/quantizer/impl/simple.py:13-55).  Runs as a HIP kernel; GPU tensors only."""

from __future__ import annotations

import torch

from ... import engine
from ...data.dtype import QuantDataType
from ...data.range import QuantRange

__all__ = ["simple_quantize"]


def simple_quantize(tensor: torch.Tensor, *, quant_dtype: torch.dtype | QuantDataType, has_zero_point: bool,
                    quant_range: QuantRange | None = None, round_delta: torch.Tensor | None = None) -> torch.Tensor:
    if not isinstance(quant_dtype, (torch.dtype, QuantDataType)):
        raise TypeError(f"quant_dtype must be either torch.dtype or QuantDataType, got {quant_dtype}")
    if round_delta is not None:
        # simple.py:22-51: torch dtypes add the offset after the cast; integer types take floor + offset
        if isinstance(quant_dtype, QuantDataType):
            assert not quant_dtype.is_exponent, "round_delta is not supported for exponential quantization"
            assert not quant_dtype.is_float_point, "round_delta is not supported for float quantization"
            qr = QuantRange.construct(quant_dtype, has_zero_point=has_zero_point, quant_range=quant_range)
            return tensor.floor().add_(round_delta).clamp_(min=qr.min, max=qr.max)
        out = tensor.to(dtype=quant_dtype).to(dtype=tensor.dtype).add_(round_delta)
        if quant_range is not None and quant_range.is_set():
            out = out.clamp(min=quant_range.min, max=quant_range.max)
        return out
    if tensor.requires_grad:
        raise NotImplementedError("the straight-through-estimator path is not implemented (calibration runs without grad)")
    rounding = engine.make_rounding(quant_dtype, has_zero_point=has_zero_point, quant_range=quant_range,
                                    device=tensor.device)
    work = tensor.to(torch.float32)
    return engine.simple_round_signed(rounding, work).to(tensor.dtype)
