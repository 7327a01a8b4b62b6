"""Per-channel statistics (reference: deepcompressor/calib/metric.py:13-195).

Same static methods as the reference's ``ChannelMetric``.  A tensor is viewed as [N, num_channels, -1] and reduced
over the first and last axes by ``dcq_channel_metric`` (one pass, HBM-bound); lists of tensors accumulate into the
same output, which is what the reference's maximum / sum over the per-tensor results amounts to.
``abs_normalize_mean`` first divides every value by the largest magnitude of its group (``dcq_group_range``).
"""

from __future__ import annotations

import ctypes as C
import typing as tp

import torch

from .._lib import DcqError, check, lib, storage_of, stream_of

__all__ = ["ChannelMetric"]

_ABS_MAX, _ABS_SUM, _SQUARE_SUM = 0, 1, 2


def _as_list(tensors) -> list[torch.Tensor]:
    return [tensors] if isinstance(tensors, torch.Tensor) else list(tensors)


def _reduce(tensors, num_channels: int, mode: int, device, dtype: torch.dtype) -> tuple[torch.Tensor, int]:
    tensors = _as_list(tensors)
    dev = torch.device(device) if device is not None else tensors[0].device
    if dev.type != "cuda":
        dev = next((t.device for t in tensors if t.is_cuda), None)
        if dev is None:
            raise DcqError("deepcompressor_amd runs on MI355X GPUs only (there is no CPU path)")
    out_max = torch.zeros(num_channels, dtype=torch.float32, device=dev) if mode == _ABS_MAX else None
    out_sum = torch.zeros(num_channels, dtype=torch.float64, device=dev) if mode != _ABS_MAX else None
    count = 0
    for t in tensors:
        t = t.to(dev, non_blocking=True).contiguous()
        n = t.shape[0]
        inner = t.numel() // max(1, n * num_channels)
        assert n * num_channels * inner == t.numel(), "tensor does not view as [N, num_channels, -1]"
        count += n * inner
        if t.numel() == 0:
            continue
        check(lib().dcq_channel_metric(C.c_void_p(t.data_ptr()), storage_of(t.dtype), n, num_channels, inner, mode,
                                       C.c_void_p(out_max.data_ptr()) if out_max is not None else None,
                                       C.c_void_p(out_sum.data_ptr()) if out_sum is not None else None,
                                       stream_of(dev)), "dcq_channel_metric")
    result = out_max if mode == _ABS_MAX else out_sum.to(torch.float32)
    out_device = torch.device(device) if device is not None else tensors[0].device
    return result.to(dtype=dtype, device=out_device if out_device.type == "cuda" else dev), count


class ChannelMetric:
    """Channel-wise metrics R[c] over tensors viewed as [N, num_channels, -1]."""

    @staticmethod
    def abs_max(tensors: tp.Iterable[torch.Tensor] | torch.Tensor, num_channels: int, group_shape: tp.Sequence[int],
                device: torch.device | str = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """max |T[:, c, :]| (metric.py:142-152); exact."""
        return _reduce(tensors, num_channels, _ABS_MAX, device, dtype)[0]

    @staticmethod
    def abs_mean(tensors: tp.Iterable[torch.Tensor] | torch.Tensor, num_channels: int, group_shape: tp.Sequence[int],
                 device: torch.device | str = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """mean |T[:, c, :]| (metric.py:155-166)."""
        total, count = _reduce(tensors, num_channels, _ABS_SUM, device, dtype)
        return total.div_(count)

    @staticmethod
    def abs_normalize_mean(tensors, num_channels: int, group_shape: tp.Sequence[int], device=None,
                           dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """mean |T[:, c, :]| / max |group| with groups of ``group_shape`` (metric.py:16-30, 169-181)."""
        from ..data.range import _group_min_max
        from ..data.utils.shape import infer_view_shape

        normalized = []
        for t in _as_list(tensors):
            if not t.is_cuda:
                t = t.to(device if device is not None and torch.device(device).type == "cuda" else "cuda")
            view = t.contiguous().view(infer_view_shape(t.shape, group_shape)).to(dtype)
            vmax, _ = _group_min_max(view, with_min=False)
            vmax = torch.where(vmax == 0, torch.ones_like(vmax), vmax).to(dtype)
            normalized.append((view / vmax).view(t.shape))
        total, count = _reduce(normalized, num_channels, _ABS_SUM, device, dtype)
        return total.div_(count)

    @staticmethod
    def root_mean_square(tensors: tp.Iterable[torch.Tensor] | torch.Tensor, num_channels: int,
                         group_shape: tp.Sequence[int], device: torch.device | str = None,
                         dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """sqrt(mean T[:, c, :]^2) (metric.py:184-195)."""
        total, count = _reduce(tensors, num_channels, _SQUARE_SUM, device, dtype)
        return total.div_(count).sqrt_()
