"""Element-wise snapping of a tensor onto a data type
(reference: deepcompressor/quantizer/impl/simple.py:13-55).  Runs as a HIP kernel; GPU tensors only."""

from __future__ import annotations

import torch

from ... import engine
from ...data.dtype import QuantDataType
from ...data.range import LogQuantRange, QuantRange

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
            floored = _FloorThrough.apply(tensor) if tensor.requires_grad else tensor.floor()
            return floored.add_(round_delta).clamp_(min=qr.min, max=qr.max)
        out = tensor.to(dtype=quant_dtype).to(dtype=tensor.dtype).add_(round_delta)
        if quant_range is not None and quant_range.is_set():
            out = out.clamp(min=quant_range.min, max=quant_range.max)
        return out
    if tensor.requires_grad:
        return _StraightThrough.apply(tensor, quant_dtype, has_zero_point, quant_range)
    return _snap(tensor, quant_dtype, has_zero_point, quant_range)


def _snap(tensor: torch.Tensor, quant_dtype, has_zero_point: bool, quant_range: QuantRange | None) -> torch.Tensor:
    rounding = engine.make_rounding(quant_dtype, has_zero_point=has_zero_point, quant_range=quant_range,
                                    device=tensor.device)
    work = tensor.to(torch.float32)
    return engine.simple_round_signed(rounding, work).to(tensor.dtype)


class _FloorThrough(torch.autograd.Function):
    """floor with the identity as its derivative (quantizer/impl/simple.py:49 through quantizer/impl/ste.py:11-27)."""

    @staticmethod
    def forward(ctx, tensor):
        return tensor.floor()

    @staticmethod
    def backward(ctx, grad):
        return grad


class _StraightThrough(torch.autograd.Function):
    """Rounding with the identity as its derivative; the clamps keep theirs (reference: quantizer/impl/ste.py:11-27
    and the clamp / clamp_ calls around it in quantizer/impl/simple.py:23-51).  The forward is the same HIP kernel as
    without grad; the backward multiplies the incoming gradient by the pass-through mask of the reference's graph:
    integers, ``clamp_(round(x))``: 1 where round(x) lies inside the range (bounds included); floats,
    ``clamp_(round(clamp(x)))``: 1 where x lies inside the type's value range and the rounded value inside
    ``quant_range``; exponents, ``exp2(clamp_(floor(log2 x)))``: q / x where the floored exponent is inside the range;
    torch dtypes, ``clamp(cast(x))``: 1 where the cast value is inside ``quant_range``, and the gradient itself is
    rounded to that dtype (the backward of a cast is a cast)."""

    @staticmethod
    def forward(ctx, tensor, quant_dtype, has_zero_point, quant_range):
        x = tensor.detach()
        out = _snap(x, quant_dtype, has_zero_point, quant_range)
        extra = quant_range is not None and quant_range.is_set()
        if isinstance(quant_dtype, torch.dtype):
            factor = _inside(x.to(quant_dtype).to(x.dtype), quant_range.min, quant_range.max) if extra else None
        elif quant_dtype.is_exponent:
            log_range = LogQuantRange.construct(quant_dtype, quant_range)
            factor = _inside(x.log2().floor_(), log_range.min, log_range.max).to(x.dtype).mul_(out).div_(x)
        elif quant_dtype.is_float_point:
            factor = _inside(x, quant_dtype.min_value, quant_dtype.max_value)
            if extra:
                factor &= _inside(_snap(x, quant_dtype, has_zero_point, None), quant_range.min, quant_range.max)
        else:
            q = QuantRange.construct(quant_dtype, has_zero_point=has_zero_point, quant_range=quant_range)
            factor = _inside(x.round(), q.min, q.max)
        ctx.factor, ctx.through = factor, quant_dtype if isinstance(quant_dtype, torch.dtype) else None
        return out

    @staticmethod
    def backward(ctx, grad):
        if ctx.factor is not None:
            grad = grad * ctx.factor.to(grad.dtype)
        if ctx.through is not None:
            grad = grad.to(ctx.through).to(grad.dtype)
        return grad, None, None, None


def _inside(values: torch.Tensor, lo, hi) -> torch.Tensor:
    mask = torch.ones_like(values, dtype=torch.bool)
    if lo is not None:
        mask &= values >= lo
    if hi is not None:
        mask &= values <= hi
    return mask
