"""Reshape callables that turn module inputs / outputs into 2-D GEMM operands [samples][channels]
(reference: deepcompressor/data/utils/reshape.py:16-160).  Pure layout work on the caller's tensors."""

import torch
import torch.nn.functional as F

__all__ = ["ReshapeFn", "LinearReshapeFn", "ConvInputReshapeFn", "ConvOutputReshapedFn", "AttentionInputReshapeFn"]


def _channels_to_matrix(x3: torch.Tensor, ic_last: bool) -> torch.Tensor:
    """[batch][channels][positions] -> [batch * positions][channels], or [channels][batch * positions]."""
    channels = x3.shape[1]
    if ic_last:
        return x3.permute(0, 2, 1).reshape(-1, channels)
    return x3.permute(1, 0, 2).reshape(channels, -1)


class ReshapeFn:
    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        return x


class LinearReshapeFn(ReshapeFn):
    """(..., C) -> (N, C), or (C, N) when ``ic_last`` is false."""

    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        flat = x.view(-1, x.shape[-1])
        return flat if ic_last else flat.t()


class ConvInputReshapeFn(ReshapeFn):
    """Inputs of a convolution as the rows its weight matrix [out][in * kernel] multiplies: one row per output
    position, the receptive field unfolded in (channel, kernel row, kernel column) order."""

    def __init__(self, kernel_size: tuple[int, ...], padding: tuple[int, ...], stride: tuple[int, ...],
                 dilation: tuple[int, ...]) -> None:
        self.kernel_size, self.padding, self.stride, self.dilation = kernel_size, padding, stride, dilation

    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        patches = F.unfold(x, kernel_size=self.kernel_size, padding=self.padding, stride=self.stride,
                           dilation=self.dilation)  # [batch][in * kernel][positions]
        return _channels_to_matrix(patches, ic_last)


class ConvOutputReshapedFn(ReshapeFn):
    """Outputs of a convolution [batch][channels][...] as [batch * positions][channels]."""

    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        return _channels_to_matrix(x.view(x.shape[0], x.shape[1], -1), ic_last)


class AttentionInputReshapeFn(ReshapeFn):
    """Attention operands whose channels sit at ``channels_dim``: everything before it is the batch, everything after
    it the positions."""

    def __init__(self, channels_dim: int) -> None:
        self.channels_dim = channels_dim

    def __call__(self, x: torch.Tensor, /, ic_last: bool = True) -> torch.Tensor:
        d = self.channels_dim
        before, after = x.shape[:d], x.shape[d + 1:]
        return _channels_to_matrix(x.view(before.numel(), x.shape[d], after.numel()), ic_last)
