"""2-D convolutions rewritten for quantization (reference: deepcompressor/nn/patch/conv.py:14-182), the convolution
counterparts of ``ConcatLinear`` / ``ShiftedLinear``: one convolution per part of a channel-concatenated input, and a
convolution whose inner input is shifted (padding included) with the bias compensating."""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["ConcatConv2d", "ShiftedConv2d"]


def _geometry(conv: nn.Conv2d) -> dict:
    return dict(kernel_size=conv.kernel_size, stride=conv.stride, padding=conv.padding, dilation=conv.dilation,
                groups=conv.groups, padding_mode=conv.padding_mode, device=conv.weight.device, dtype=conv.weight.dtype)


class ConcatConv2d(nn.Module):
    def __init__(self, in_channels_list: list[int], out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = "zeros", device=None, dtype=None) -> None:
        super().__init__()
        assert len(in_channels_list) > 1, "ConcatConv2d requires at least 2 input channels"
        self.in_channels_list = in_channels_list
        self.in_channels = sum(in_channels_list)
        last = len(in_channels_list) - 1
        self.convs = nn.ModuleList(
            nn.Conv2d(width, out_channels, kernel_size, stride, padding, dilation, groups, bias and i == last,
                      padding_mode, device, dtype) for i, width in enumerate(in_channels_list))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        total = None
        for conv, part in zip(self.convs, x.split(self.in_channels_list, dim=1), strict=True):
            y = conv(part.contiguous())  # contiguous: the part is what the input quantizer sees
            total = y if total is None else total + y
        return total

    @staticmethod
    def from_conv2d(conv: nn.Conv2d, splits: list[int]) -> "ConcatConv2d":
        """``splits``: widths of the leading parts; the remaining channels form the last part (the list is extended
        in place, as the reference does)."""
        splits.append(conv.in_channels - sum(splits))
        widths = [w for w in splits if w > 0]
        assert len(widths) > 1, "ConcatConv2d requires at least 2 input channels"
        out = ConcatConv2d(widths, conv.out_channels, bias=conv.bias is not None, **_geometry(conv))
        begin = 0
        for part in out.convs:
            part.weight.data.copy_(conv.weight[:, begin:begin + part.in_channels])
            begin += part.in_channels
        if conv.bias is not None:
            out.convs[-1].bias.data.copy_(conv.bias)
        return out


class ShiftedConv2d(nn.Module):
    shift: torch.Tensor

    def __init__(self, in_channels: int, out_channels: int, kernel_size, shift: float | torch.Tensor, stride=1,
                 padding=0, dilation=1, groups: int = 1, bias: bool = True, padding_mode: str = "zeros", device=None,
                 dtype=None) -> None:
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                              padding_mode, device, dtype)
        self.conv.shifted = True
        shift = torch.as_tensor(shift, device=device, dtype=dtype).flatten().to(device=device, dtype=dtype)
        if shift.numel() > 1:  # one value per block of consecutive input channels
            unpadded = padding == 0 or (isinstance(padding, (tuple, list)) and all(p == 0 for p in padding))
            assert unpadded, "Padding is not supported for multi-channel shift"
            assert in_channels >= shift.numel() and in_channels % shift.numel() == 0
            shift = shift.repeat_interleave(in_channels // shift.numel())
        self.register_buffer("shift", shift.view(1, -1, 1, 1))
        # the module pads the *shifted* input itself (zero padding becomes padding with the shift value, so that the
        # padded border still stands for x = 0) and the inner convolution runs unpadded
        self.padding_size = self.conv._reversed_padding_repeated_twice
        self.padding_mode, self.padding_value = self.conv.padding_mode, None
        if all(p == 0 for p in self.padding_size):
            self.padding_mode = ""
        elif self.padding_mode == "zeros":
            assert shift.numel() == 1, "Zero padding is not supported for multi-channel shift"
            self.padding_mode, self.padding_value = "constant", shift.item()
        self.conv.padding, self.conv.padding_mode = "valid", "zeros"
        self.conv._reversed_padding_repeated_twice = [0, 0] * len(self.conv.kernel_size)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        input = input + self.shift
        if self.padding_mode:
            input = F.pad(input, self.padding_size, mode=self.padding_mode, value=self.padding_value)
        return self.conv(input)

    @staticmethod
    def from_conv2d(conv: nn.Conv2d, shift: float | torch.Tensor) -> "ShiftedConv2d":
        """bias' = bias - sum over the kernel of W times the shift (float64), so that the module computes what ``conv``
        did."""
        geometry = _geometry(conv)
        out = ShiftedConv2d(conv.in_channels, conv.out_channels, shift=shift, bias=True, **geometry)
        out.conv.weight.data.copy_(conv.weight)
        w64, s64 = conv.weight.double(), out.shift.double().view(-1)
        moved = w64.sum(dim=[1, 2, 3]) * s64 if s64.numel() == 1 else w64.sum(dim=[2, 3]) @ s64
        base = conv.bias.data.double() if conv.bias is not None else torch.zeros_like(moved)
        out.conv.bias.data.copy_((base - moved).to(geometry["dtype"]))
        return out
