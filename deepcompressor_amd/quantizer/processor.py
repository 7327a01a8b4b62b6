"""``Quantizer``: the drop-in entry point of the fake-quantization path
(reference: deepcompressor/quantizer/processor.py:23-366).

Same constructor fields, ``quantize`` keyword arguments (``MISSING`` falls back to the field), ``update``,
``state_dict`` / ``load_state_dict`` layout and ``process`` / ``as_hook`` helpers as the reference; the work is
done by the HIP kernels behind :class:`QuantizerImpl`.
"""

from __future__ import annotations

import typing as tp
from dataclasses import _MISSING_TYPE, MISSING, dataclass

import torch
import torch.nn as nn

from ..data.range import DynamicRange, QuantRange, RangeBound
from ..data.tensor import QuantTensor
from ..utils.common import tree_map
from ..utils.config import KeyEnableConfig
from .config.kernel import BaseKeyEnableQuantKernelConfig, BaseQuantKernel, BaseQuantKernelConfig
from .config.lowrank import QuantLowRankConfig
from .impl.base import QuantizerImpl
from .impl.info import QuantInfo

__all__ = ["Quantizer", "ProcessHook"]


class ProcessHook:
    """Forward (pre-)hook that passes a module's input or output tensors through a processor."""

    def __init__(self, processor: "Quantizer", func=None, is_output: bool = False):
        self.processor, self.func, self.is_output = processor, func, is_output
        self.input_packager = processor.get_input_packager()
        self.output_packager = processor.get_output_packager()
        self._handles: list = []

    def _apply(self, tensors: dict) -> dict:
        fn = self.func if self.func is not None else self.processor.process
        return {k: fn(v) for k, v in tensors.items()}

    def _pre(self, module, args, kwargs):
        if not self.processor.is_enabled():
            return None
        if self.input_packager is not None:
            return self.input_packager.repack(self._apply(self.input_packager.unpack(module, args, kwargs)), module,
                                              args, kwargs)
        return (self._apply({0: args[0]})[0], *args[1:]), kwargs

    def _post(self, module, args, kwargs, output):
        if not self.processor.is_enabled():
            return None
        if self.output_packager is not None:
            return self.output_packager.repack(
                self._apply(self.output_packager.unpack(module, args, kwargs, output)), module, args, kwargs, output)
        if isinstance(output, torch.Tensor):
            return self._apply({0: output})[0]
        return (self._apply({0: output[0]})[0], *output[1:])

    def register(self, module: nn.Module | tp.Iterable[nn.Module]) -> "ProcessHook":
        modules = [module] if isinstance(module, nn.Module) else list(module)
        for m in modules:
            if self.is_output:
                self._handles.append(m.register_forward_hook(self._post, with_kwargs=True))
            else:
                self._handles.append(m.register_forward_pre_hook(self._pre, with_kwargs=True))
        return self

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles.clear()


@dataclass
class Quantizer(QuantizerImpl):
    kernel: BaseKeyEnableQuantKernelConfig | BaseQuantKernelConfig | BaseQuantKernel | None = None
    channels_dim: int | None = None
    scale: torch.Tensor | tp.Sequence[torch.Tensor] | None = None
    zero: torch.Tensor | None = None
    dynamic_range: DynamicRange | tp.Sequence[DynamicRange] | None = None
    range_bound: RangeBound | None = None
    quant_range: QuantRange | None = None
    default_dtype: torch.dtype | None = None
    develop_dtype: torch.dtype = torch.float32
    low_rank: QuantLowRankConfig | None = None
    input_packager: tp.Any = None
    output_packager: tp.Any = None

    def is_enabled_low_rank(self) -> bool:
        if self.low_rank is None:
            return False
        if isinstance(self.low_rank, KeyEnableConfig):
            return self.low_rank.is_enabled_for(self.key)
        return self.low_rank.is_enabled()

    def get_input_packager(self):
        return self.input_packager

    def get_output_packager(self):
        return self.output_packager

    def process(self, tensor: torch.Tensor) -> torch.Tensor:
        return self.quantize(tensor).data

    def as_hook(self, func=None, *, is_output: bool = False) -> ProcessHook:
        return ProcessHook(self, func, is_output=is_output)

    def quantize(self, tensor: torch.Tensor, /, *, return_with_dequant: bool = True, return_with_quant: bool = False,
                 kernel=MISSING, channels_dim=MISSING, scale=MISSING, zero=MISSING, dynamic_range=MISSING,
                 range_bound=MISSING, quant_range=MISSING, default_dtype=MISSING, develop_dtype=MISSING,
                 **kwargs) -> QuantTensor:
        pick = lambda given, own: own if given is MISSING else given  # noqa: E731
        kernel = pick(kernel, self.kernel)
        if isinstance(kernel, BaseKeyEnableQuantKernelConfig):
            kernel = kernel.specialize_for(self.key)
        elif isinstance(kernel, KeyEnableConfig):
            kernel = kernel if kernel.is_enabled_for(self.key) else None
        assert isinstance(kernel, (BaseQuantKernel, BaseQuantKernelConfig, type(None)))
        return super().quantize(
            tensor, kernel=kernel, channels_dim=pick(channels_dim, self.channels_dim), scale=pick(scale, self.scale),
            zero=pick(zero, self.zero), dynamic_range=pick(dynamic_range, self.dynamic_range),
            range_bound=pick(range_bound, self.range_bound), quant_range=pick(quant_range, self.quant_range),
            return_with_dequant=return_with_dequant, return_with_quant=return_with_quant,
            default_dtype=pick(default_dtype, self.default_dtype),
            develop_dtype=pick(develop_dtype, self.develop_dtype), **kwargs)

    def smooth_candidates(self, tensor: torch.Tensor, chan_scale: torch.Tensor, *, divide_first: bool,
                          channels_dim=MISSING, default_dtype=MISSING, errors_degree: float | None = None,
                          as_codes: bool = False) -> torch.Tensor | tuple[torch.Tensor, torch.Tensor] | None:
        """``QuantizerImpl.smooth_candidates`` with this quantizer's own scale / zero / ranges (round-to-nearest: the
        searches never use another kernel, calib/smooth.py:526-528).  None = not covered by the fused kernel."""
        pick = lambda given, own: own if given is MISSING else given  # noqa: E731
        if not self.is_enabled():
            return None
        return super().smooth_candidates(
            tensor, chan_scale, divide_first=divide_first, channels_dim=pick(channels_dim, self.channels_dim),
            scale=self.scale, zero=self.zero, dynamic_range=self.dynamic_range, range_bound=self.range_bound,
            quant_range=self.quant_range, default_dtype=pick(default_dtype, self.default_dtype),
            errors_degree=errors_degree, as_codes=as_codes)

    def update(self, tensor_shape: torch.Size, default_dtype: torch.dtype | _MISSING_TYPE = MISSING,
               quant_range: QuantRange | None | _MISSING_TYPE = MISSING,
               range_bound: RangeBound | None | _MISSING_TYPE = MISSING) -> QuantInfo | None:
        return super().update(
            tensor_shape,
            default_dtype=self.default_dtype if default_dtype is MISSING else default_dtype,
            quant_range=self.quant_range if quant_range is MISSING else quant_range,
            range_bound=self.range_bound if range_bound is MISSING else range_bound)

    # ---- checkpoint layout (wgts.pt / acts.pt entries) ---------------------------------------
    def quantize_with_low_rank(self, tensors: torch.Tensor | tp.Sequence[torch.Tensor], /, *,
                               return_with_dequant: bool = True, return_with_quant: bool = False, **qkwargs):
        """Quantize weights together with their low-rank branch (reference: quantizer/processor.py:215-310).

        One branch per tensor (single tensor or ``exclusive``) or one branch fitted on the row-wise concatenation
        with the up-projection split back per tensor.  ``compensate``: quantize first, fit the branch to the
        error; otherwise fit the branch to the weight and quantize the remainder.  Returns (QuantTensors, branches)
        -- branches is None when the low-rank branch is disabled."""
        from ..nn.patch.lowrank import LowRankBranch

        tensors = [tensors] if isinstance(tensors, torch.Tensor) else list(tensors)
        qkwargs = dict(return_with_dequant=return_with_dequant, return_with_quant=return_with_quant, **qkwargs)
        if not self.is_enabled_low_rank():
            return [self.quantize(t.data, **qkwargs) for t in tensors], None
        rank, compensate = self.low_rank.rank, self.low_rank.compensate
        if compensate:
            qkwargs["return_with_dequant"] = True
        qtensors, branches = [], []
        if len(tensors) == 1 or self.low_rank.exclusive:
            for t in tensors:
                if compensate:
                    qt = self.quantize(t.data, **qkwargs)
                    lb = LowRankBranch(t.shape[1], t.shape[0], rank=rank, weight=t.data - qt.data)
                else:
                    lb = LowRankBranch(t.shape[1], t.shape[0], rank=rank, weight=t.data)
                    qt = self.quantize(t.data - lb.get_effective_weight().view(t.data.shape), **qkwargs)
                qtensors.append(qt)
                branches.append(lb)
            return qtensors, branches
        stacked = torch.cat([t.data for t in tensors], dim=0)
        if compensate:
            qtensors = [self.quantize(t.data, **qkwargs) for t in tensors]
            stacked = stacked - torch.cat([q.data for q in qtensors], dim=0)
        shared = LowRankBranch(stacked.shape[1], stacked.shape[0], rank=rank, weight=stacked)
        del stacked
        start = 0
        for t in tensors:
            lb = LowRankBranch(t.shape[1], t.shape[0], rank=rank)
            lb.a = shared.a
            lb.b.to(dtype=t.dtype, device=t.device)
            lb.b.weight.data.copy_(shared.b.weight[start:start + t.shape[0]])
            start += t.shape[0]
            if not compensate:
                qtensors.append(self.quantize(t.data - lb.get_effective_weight(), **qkwargs))
            branches.append(lb)
        return qtensors, branches

    def state_dict(self, device: torch.device | str = "cpu", non_blocking: bool = False) -> dict[str, tp.Any]:
        """quantizer/processor.py:312-339.  ``non_blocking`` (no counterpart in the reference): device tensors bound
        for the host are copied asynchronously into page-locked memory on the current stream; the caller synchronises
        that stream once before reading any of them, instead of once per tensor."""
        to_pinned = non_blocking and torch.device(device).type == "cpu"

        def copy(t: torch.Tensor) -> torch.Tensor:
            if to_pinned and t.is_cuda:
                host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                return host.copy_(t, non_blocking=True)
            return t.to(device).clone()

        if self.dynamic_range is None:
            dyn = None
        elif isinstance(self.dynamic_range, DynamicRange):
            dyn = tree_map(copy, self.dynamic_range.to_dict())
        else:
            dyn = tree_map(copy, tuple(d.to_dict() for d in self.dynamic_range))
        return {
            "channels_dim": self.channels_dim,
            "scale": tree_map(copy, self.scale),
            "zero": copy(self.zero) if self.zero is not None else None,
            "dynamic_range": dyn,
            "range_bound": self.range_bound.to_dict() if self.range_bound is not None else None,
            "quant_range": self.quant_range.to_dict() if self.quant_range is not None else None,
        }

    def load_state_dict(self, state_dict: dict[str, tp.Any], device: torch.device | str = "cpu"):
        move = lambda t: t.to(device)  # noqa: E731
        self.channels_dim = state_dict["channels_dim"]
        self.scale = tree_map(move, state_dict["scale"])
        self.zero = move(state_dict["zero"]) if state_dict["zero"] is not None else None
        dyn = state_dict["dynamic_range"]
        if dyn is None:
            self.dynamic_range = None
        elif isinstance(dyn, dict):
            self.dynamic_range = DynamicRange.from_dict(tree_map(move, dyn))
        else:
            self.dynamic_range = tuple(DynamicRange.from_dict(tree_map(move, d)) for d in dyn)
        self.range_bound = RangeBound.from_dict(state_dict["range_bound"])
        self.quant_range = QuantRange.from_dict(state_dict["quant_range"])
