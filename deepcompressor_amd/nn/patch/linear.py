"""Linear layers rewritten for quantization (reference: deepcompressor/nn/patch/linear.py:10-128).

``ConcatLinear``: a linear whose input is a concatenation, as one linear per part (each part gets its own
quantizer, smoothing scale and low-rank branch); the parts' outputs are added, the bias sits on the last part.
``ShiftedLinear``: the inner linear sees ``x + shift`` and its bias compensates, so the module computes what the
original linear did while the inner input (now non-negative) can take an unsigned activation type.  These are plain modules of the model being
calibrated: they run wherever the model does.
"""

from __future__ import annotations

import torch
import torch.nn as nn

__all__ = ["ConcatLinear", "ShiftedLinear"]


class ConcatLinear(nn.Module):
    def __init__(self, in_features_list: list[int], out_features: int, bias: bool = True, device=None,
                 dtype=None) -> None:
        super().__init__()
        assert len(in_features_list) > 1, "ConcatLinear requires at least 2 input features"
        self.in_features_list = list(in_features_list)
        self.in_features, self.out_features = sum(self.in_features_list), out_features
        last = len(self.in_features_list) - 1
        self.linears = nn.ModuleList(nn.Linear(width, out_features, bias and i == last, device, dtype)
                                     for i, width in enumerate(self.in_features_list))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        total = None
        for linear, part in zip(self.linears, x.split(self.in_features_list, dim=-1), strict=True):
            y = linear(part.contiguous())  # contiguous: the part is what the input quantizer sees
            total = y if total is None else total + y
        return total

    @staticmethod
    def from_linear(linear: nn.Linear, splits: list[int]) -> "ConcatLinear":
        """``splits``: widths of the leading parts; what is left of the input forms the last part.  The list is
        extended in place, as the reference does."""
        splits.append(linear.in_features - sum(splits))
        widths = [w for w in splits if w > 0]
        assert len(widths) > 1, "ConcatLinear requires at least 2 input features"
        out = ConcatLinear(widths, linear.out_features, bias=linear.bias is not None, device=linear.weight.device,
                           dtype=linear.weight.dtype)
        begin = 0
        for part in out.linears:
            part.weight.data.copy_(linear.weight[:, begin:begin + part.in_features])
            begin += part.in_features
        if linear.bias is not None:
            out.linears[-1].bias.data.copy_(linear.bias)
        return out


class ShiftedLinear(nn.Module):
    shift: torch.Tensor

    def __init__(self, in_features: int, out_features: int, shift: float | torch.Tensor, bias: bool = True,
                 device=None, dtype=None) -> None:
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias, device, dtype)
        self.linear.shifted = True
        weight = self.linear.weight
        shift = torch.as_tensor(shift, device=weight.device, dtype=weight.dtype).flatten()
        if shift.numel() > 1:  # one value per block of consecutive input channels
            assert in_features >= shift.numel() and in_features % shift.numel() == 0
            shift = shift.repeat_interleave(in_features // shift.numel())
        self.register_buffer("shift", shift)

    @property
    def in_features(self) -> int:
        return self.linear.in_features

    @property
    def out_features(self) -> int:
        return self.linear.out_features

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self.linear(input + self.shift.view([1] * (input.dim() - 1) + [-1]))

    @staticmethod
    def from_linear(linear: nn.Linear, shift: float | torch.Tensor) -> "ShiftedLinear":
        """bias' = bias - W shift (float64), so that ``shifted(x) == linear(x)``.  (With one shift per channel the
        reference multiplies W [out, in] by a [1, in] row, which cannot run; the product meant is W shift.)"""
        weight = linear.weight
        out = ShiftedLinear(linear.in_features, linear.out_features, shift, bias=True, device=weight.device,
                            dtype=weight.dtype)
        out.linear.weight.data.copy_(weight)
        w64, s64 = weight.double(), out.shift.double()
        moved = w64.sum(dim=1) * s64 if s64.numel() == 1 else w64 @ s64
        base = linear.bias.data.double() if linear.bias is not None else torch.zeros_like(moved)
        out.linear.bias.data.copy_((base - moved).to(weight.dtype))
        return out
