"""Ranges: bounds, quantization ranges and dynamic (data) ranges.

API of the reference's deepcompressor/data/range.py:16-436.  The only O(numel) work in here, the
per-group min/max of ``DynamicRange.measure``, is the HIP kernel ``dcq_group_range``; everything else
is bookkeeping on per-group tensors.
"""

from __future__ import annotations

import math
import typing as tp
from dataclasses import dataclass

import numpy as np
import torch

from .dtype import QuantDataType
from .zero import ZeroPointDomain

__all__ = ["RangeBound", "QuantRange", "LogQuantRange", "ProtectiveQuantRange", "DynamicRange"]


@dataclass
class RangeBound:
    min: float | None = None
    max: float | None = None

    def is_set(self) -> bool:
        return self.min is not None or self.max is not None

    def to_dict(self) -> dict[str, tp.Any]:
        return {"min": self.min, "max": self.max}

    @classmethod
    def from_dict(cls, data: dict[str, tp.Any] | None):
        return None if data is None else cls(min=data["min"], max=data["max"])


class QuantRange(RangeBound):
    """Range of the quantized values, e.g. [-7, 7] for sint4 without zero point."""

    def log2(self) -> "LogQuantRange":
        magnitude = min(abs(self.min or 0), abs(self.max or 0))
        return LogQuantRange(min=None, max=None if self.max is None else int(math.log2(magnitude)))

    def intersect(self, quant_dtype: QuantDataType, *, has_zero_point: bool) -> "QuantRange":
        hi = quant_dtype.max_value if self.max is None else min(self.max, quant_dtype.max_value)
        lo = quant_dtype.min_value if self.min is None else max(self.min, quant_dtype.min_value)
        if quant_dtype.signed and not has_zero_point:  # symmetric: sint4 -> [-7, 7]
            hi = min(abs(lo), abs(hi))
            lo = -hi
        return QuantRange(min=lo, max=hi)

    def intersect_log2(self, quant_dtype: QuantDataType) -> "LogQuantRange":
        return self.log2().intersect_log2(quant_dtype)

    @staticmethod
    def construct(dtype: QuantDataType, *, has_zero_point: bool,
                  quant_range: tp.Optional["QuantRange"] = None) -> "QuantRange":
        return (quant_range or QuantRange()).intersect(dtype, has_zero_point=has_zero_point)


class LogQuantRange(QuantRange):
    """Range of exponents for exponent-only data types."""

    def log2(self) -> "LogQuantRange":
        return self

    def intersect(self, quant_dtype: QuantDataType, *, has_zero_point: bool) -> "QuantRange":
        raise NotImplementedError("LogQuantRange does not support intersect method")

    def intersect_log2(self, quant_dtype: QuantDataType) -> "LogQuantRange":
        hi = quant_dtype.max_exponent_value if self.max is None else min(self.max, quant_dtype.max_exponent_value)
        lo = quant_dtype.min_exponent_value if self.min is None else max(self.min, quant_dtype.min_exponent_value)
        return LogQuantRange(min=lo, max=hi)

    @staticmethod
    def construct(dtype: QuantDataType,
                  quant_range: tp.Optional[tp.Union["LogQuantRange", QuantRange]] = None) -> "LogQuantRange":
        return (quant_range or LogQuantRange()).intersect_log2(dtype)


class ProtectiveQuantRange(QuantRange):
    """Largest symmetric range [-p, p] of an outer integer type such that re-quantizing any sub-range of it
    with the inner (zero-point) integer type and de-quantizing never leaves the outer type's range."""

    _instances: tp.ClassVar[dict] = {}

    @staticmethod
    def construct(outer_dtype: QuantDataType, inner_dtype: QuantDataType, zero_domain: ZeroPointDomain | None,
                  inner_quant_range: QuantRange | None = None) -> QuantRange:
        assert outer_dtype.is_integer and inner_dtype.is_integer, "both levels must be integer types"
        assert zero_domain is not None or outer_dtype.signed == inner_dtype.signed
        if zero_domain is None:
            return QuantRange.construct(outer_dtype, has_zero_point=False)
        inner = QuantRange.construct(inner_dtype, has_zero_point=True, quant_range=inner_quant_range)
        qmin, qmax = int(inner.min), int(inner.max)
        key = (outer_dtype, inner_dtype, (qmin, qmax), zero_domain)
        if key not in ProtectiveQuantRange._instances:
            outer = QuantRange.construct(outer_dtype, has_zero_point=False)
            omin, omax = int(outer.min), int(outer.max)
            dmin, dmax = int(inner_dtype.min_value), int(inner_dtype.max_value)
            # every candidate value range (vmin <= vmax, vmax >= 0) at once, in float64 with half-to-even
            # rounding, which is what the scalar python arithmetic of the definition uses
            vmax = np.arange(0, omax + 1, dtype=np.float64)[None, :]
            vmin = np.arange(omin, omax + 1, dtype=np.float64)[:, None]
            s = np.rint((vmax - vmin) / (qmax - qmin))
            s = np.minimum(np.where(s == 0, 1.0, s), omax)
            s = np.where(s > 0, s, 1.0)  # cells with vmin > vmax are masked out below
            if zero_domain == ZeroPointDomain.PreScale:
                z = np.clip(np.rint(qmin - vmin / s), dmin, dmax)
                top = (np.clip(np.rint(vmax / s + z), qmin, qmax) - z) * s
                bot = (np.clip(np.rint(vmin / s + z), qmin, qmax) - z) * s
            elif zero_domain == ZeroPointDomain.PostScale:
                z = np.clip(np.rint(qmin * s - vmin), omin, omax)
                top = np.clip(np.rint((vmax + z) / s), qmin, qmax) * s - z
                bot = np.clip(np.rint((vmin + z) / s), qmin, qmax) * s - z
            else:
                raise ValueError(f"unsupported zero-point domain {zero_domain}")
            safe = (top >= omin) & (top <= omax) & (bot >= omin) & (bot <= omax)
            found = None
            for p in range(omax, 0, -1):
                rows = np.arange(-p, p + 1) - omin  # vmin in [-p, p]
                block = safe[rows[:, None], np.arange(0, p + 1)[None, :]]
                valid_cell = (np.arange(-p, p + 1)[:, None] <= np.arange(0, p + 1)[None, :])
                if bool(np.all(block | ~valid_cell)):
                    found = p
                    break
            assert found is not None, "failed to find the protective quantization range"
            ProtectiveQuantRange._instances[key] = ProtectiveQuantRange(min=-found, max=found)
        return ProtectiveQuantRange._instances[key]


def _group_min_max(view: torch.Tensor, with_min: bool) -> tuple[torch.Tensor, torch.Tensor | None]:
    """(max|x| , None) or (max, min) over the odd dims of a (#g0, gs0, #g1, gs1, ...) view, via HIP."""
    from ..engine import canonical_groups, group_range  # deferred: engine imports this module

    x2d, _ = canonical_groups(view, tuple(view.shape))
    keep = [s if i % 2 == 0 else 1 for i, s in enumerate(view.shape)]
    vmax, vmin = group_range(x2d, with_min=with_min)
    return vmax.view(keep), (vmin.view(keep) if vmin is not None else None)


@dataclass
class DynamicRange:
    """Per-group value range of the data: explicit (min, max) tensors and / or a shrink ratio."""

    min: torch.Tensor | None = None
    max: torch.Tensor | None = None
    ratio: float | torch.Tensor | None = None

    def __post_init__(self) -> None:
        if self.max is None:
            assert self.min is None, "min must be None if max is None"

    def is_set(self) -> bool:
        return self.min is not None or self.max is not None or self.ratio is not None

    def intersect(self, range_bound: RangeBound | None) -> "DynamicRange":
        assert self.max is not None, "max must be specified"
        vmax, vmin = self.max, self.min
        if range_bound is not None:
            if range_bound.max is not None:
                vmax = vmax.clamp(max=range_bound.max)
            if vmin is not None and range_bound.min is not None:
                vmin = vmin.clamp(min=range_bound.min)
        return DynamicRange(min=vmin, max=vmax)

    def measure(self, tensors: torch.Tensor | list[torch.Tensor], /, *, zero_domain: ZeroPointDomain | None,
                is_float_point: bool) -> "DynamicRange":
        """Range of tensors viewed as (#g0, gs0, #g1, gs1, ...), then ratio / sign clamp / bounds."""
        if isinstance(tensors, torch.Tensor):
            tensors = [tensors]
        first = tensors[0]
        keep = torch.Size([s if i % 2 == 0 else 1 for i, s in enumerate(first.shape)])
        if self.ratio is None and self.max is not None:  # static range
            return DynamicRange(min=self._format_m_(self.min, shape=keep, dtype=first.dtype, device=first.device),
                                max=self._format_m_(self.max, shape=keep, dtype=first.dtype, device=first.device))
        vmax = vmin = vavg = None
        centred = zero_domain is not None and is_float_point
        for t in tensors:
            tmax, tmin = _group_min_max(t, with_min=zero_domain is not None)
            vmax = tmax if vmax is None else torch.maximum(vmax, tmax.to(vmax.device))
            if tmin is not None:
                vmin = tmin if vmin is None else torch.minimum(vmin, tmin.to(vmin.device))
            if centred:  # the zero point sits at the mean of the data (data/range.py:323-328)
                tavg = t.mean(dim=list(range(1, t.ndim, 2)), keepdim=True).to(vmax.device)
                vavg = tavg if vavg is None else vavg + tavg
        if centred:
            if len(tensors) > 1:
                vavg = vavg / len(tensors)
            vmag = torch.maximum(vmax - vavg, vavg - vmin)
            if self.ratio is not None:
                vmag = vmag * self.ratio
            vmax, vmin = vavg + vmag, vavg - vmag
        elif self.ratio is not None:
            vmax = vmax * self.ratio
            if vmin is not None:
                vmin = vmin * self.ratio
        if zero_domain == ZeroPointDomain.PreScale:
            vmax, vmin = vmax.clamp(min=0), vmin.clamp(max=0)
        if self.max is not None:
            vmax = vmax.clamp(max=self.max.to(vmax.device))
            if vmin is not None and self.min is not None:
                vmin = vmin.clamp(min=self.min.to(vmin.device))
        return DynamicRange(min=vmin, max=vmax)

    def scale(self, ratio: float | torch.Tensor, zero_domain: ZeroPointDomain | None,
              is_float_point: bool) -> "DynamicRange":
        """A search candidate: this range shrunk / expanded by ``ratio``."""
        assert ratio is not None, "ratio must be specified"
        assert self.max is not None, "self.max must be specified"
        if zero_domain is None:
            assert self.min is None, "self.min must be None for data type without zero-point"
            return DynamicRange(min=None, max=self.max * ratio)
        assert self.min is not None, "self.min must be specified"
        if is_float_point:
            centre = (self.min + self.max) / 2
            half = (self.max - centre) * ratio
            lo, hi = centre - half, centre + half
        else:
            lo, hi = self.min * ratio, self.max * ratio
        if zero_domain == ZeroPointDomain.PreScale:
            lo, hi = lo.clamp(max=0), hi.clamp(min=0)
        return DynamicRange(min=lo, max=hi)

    @staticmethod
    def construct(tensors: torch.Tensor | list[torch.Tensor], /, *, zero_domain: ZeroPointDomain | None,
                  is_float_point: bool) -> "DynamicRange":
        return DynamicRange().measure(tensors, zero_domain=zero_domain, is_float_point=is_float_point)

    @staticmethod
    def _format_m_(value, *, shape: torch.Size, dtype: torch.dtype, device: torch.device) -> torch.Tensor | None:
        if value is None:
            return None
        if not isinstance(value, torch.Tensor):
            return torch.full(shape, value, dtype=dtype, device=device)
        if value.numel() == 1:
            return value.view(-1).to(dtype=dtype, device=device).expand(shape)
        if value.numel() == shape.numel():
            return value.view(shape).to(dtype=dtype, device=device)
        if value.shape[1:] == shape[1:] and value.shape[0] == 1:
            return value.to(dtype=dtype, device=device).expand(shape)
        raise ValueError(f"Invalid value shape: {value.shape}")

    def to_dict(self) -> dict[str, tp.Any]:
        return {"min": self.min, "max": self.max, "ratio": self.ratio}

    @classmethod
    def from_dict(cls, data: dict[str, tp.Any] | None):
        return None if data is None else cls(min=data["min"], max=data["max"], ratio=data["ratio"])
