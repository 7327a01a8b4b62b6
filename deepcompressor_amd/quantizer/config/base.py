"""Quantizer configurations (reference: deepcompressor/quantizer/config/base.py:25-384).

Field names, defaults, decomposition into steps and the generated cache-directory names are the
reference's, so YAML files and cache layouts written for it keep working.
"""

from __future__ import annotations

import typing as tp
from abc import abstractmethod
from dataclasses import dataclass, field

import torch

from ...data.dtype import QuantDataType
from ...data.utils import DtypeUtils, ScaleUtils, ShapeUtils
from ...data.zero import ZeroPointDomain
from ...utils.config import EnableConfig, configclass

__all__ = ["BaseQuantizerConfig", "DecomposedQuantizerConfig", "QuantizerConfig", "ProgressiveQuantizerConfig"]


class BaseQuantizerConfig(EnableConfig):
    @property
    @abstractmethod
    def quant_dtype(self) -> QuantDataType | None: ...

    @property
    @abstractmethod
    def zero_domain(self) -> ZeroPointDomain | None: ...

    @property
    @abstractmethod
    def largest_group_shape(self) -> tp.Sequence[int]: ...

    @property
    @abstractmethod
    def smallest_group_shape(self) -> tp.Sequence[int]: ...

    def is_enabled(self) -> bool:
        return self.quant_dtype is not None

    @abstractmethod
    def decompose(self) -> "DecomposedQuantizerConfig": ...

    def generate_dirnames(self, *, prefix: str = "", shape=(4096, 4096), default_dtype: torch.dtype = torch.float16,
                          **kwargs) -> list[str]:
        return self.decompose().generate_dirnames(prefix=prefix, shape=torch.Size(shape), default_dtype=default_dtype,
                                                  **kwargs)


@dataclass(frozen=True)
class DecomposedQuantizerConfig(BaseQuantizerConfig):
    """A quantizer as a sequence of single-dtype steps (one step unless progressive)."""

    steps: tuple["QuantizerConfig", ...]
    needs_dequant_saturation: bool = False

    quant_dtype = property(lambda self: self.steps[-1].dtype if self.steps else None)
    zero_domain = property(lambda self: self.steps[-1].zero_point if self.steps else None)
    largest_group_shape = property(lambda self: self.steps[0].largest_group_shape if self.steps else (-1, -1, -1))
    smallest_group_shape = property(lambda self: self.steps[-1].smallest_group_shape if self.steps else (-1, -1, -1))
    num_steps = property(lambda self: len(self.steps))

    def decompose(self) -> "DecomposedQuantizerConfig":
        return self

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, DecomposedQuantizerConfig) or self.num_steps != other.num_steps:
            return False
        same = all(a.dtype == b.dtype and a.group_shapes == b.group_shapes and a.scale_dtypes == b.scale_dtypes
                   for a, b in zip(self.steps, other.steps))
        return same and (self.num_steps <= 1 or self.needs_dequant_saturation == other.needs_dequant_saturation)

    def _get_effective_bits(self, *, shape=(4096, 4096), default_dtype: torch.dtype = torch.float16) -> float:
        shape = torch.Size(shape)
        if self.quant_dtype is None:
            return DtypeUtils.infer_dtype_bits(default_dtype)
        bits = self.quant_dtype.total_bits
        g_shape = s_dtype = None
        for step in self.steps:
            g_shapes = ShapeUtils.infer_group_shapes(step.group_shapes, shape=shape)
            s_dtypes = ScaleUtils.infer_scale_dtypes(step.scale_dtypes, default_dtype=default_dtype)
            for g_shape, s_dtype in zip(g_shapes, s_dtypes, strict=True):
                bits += DtypeUtils.infer_dtype_bits(s_dtype) / g_shape.numel()
        if self.zero_domain == ZeroPointDomain.PreScale:
            bits += self.quant_dtype.total_bits / g_shape.numel()
        elif self.zero_domain == ZeroPointDomain.PostScale:
            bits += DtypeUtils.infer_dtype_bits(s_dtype) / g_shape.numel()
        return bits

    def _get_dtype_name(self, default_dtype: torch.dtype = torch.float16) -> str:
        if self.quant_dtype is None:
            return DtypeUtils.infer_dtype_name(default_dtype)
        suffix = {ZeroPointDomain.PreScale: ".z", ZeroPointDomain.PostScale: ".zp"}.get(self.zero_domain, "")
        return DtypeUtils.infer_dtype_name(self.quant_dtype) + suffix

    def _get_group_shapes_name(self, default_dtype: torch.dtype = torch.float16) -> str:
        if self.quant_dtype is None:
            return f"tnsr.{DtypeUtils.infer_dtype_name(default_dtype)}"
        names, n, step_default = [], len(self.steps), default_dtype
        for i, step in enumerate(self.steps):
            parts = [f"{ShapeUtils.infer_group_shape_name(gs)}.{DtypeUtils.infer_dtype_name(sd or step_default)}"
                     for gs, sd in zip(step.group_shapes, step.scale_dtypes, strict=True)]
            joined = ".".join(reversed(parts))
            names.append(f"[{joined}]" if i < n - 2 else joined)
            step_default = step.dtype
            assert step_default is not None
        return ".".join(reversed(names))

    def generate_dirnames(self, *, prefix: str = "", shape=(4096, 4096), default_dtype: torch.dtype = torch.float16,
                          **kwargs) -> list[str]:
        shape = torch.Size(shape)
        names = [str(int(self._get_effective_bits(shape=shape, default_dtype=default_dtype))),
                 self._get_dtype_name(default_dtype=default_dtype),
                 self._get_group_shapes_name(default_dtype=default_dtype)]
        return [f"{prefix}.{n}" for n in names] if prefix else names


@configclass
@dataclass
class QuantizerConfig(BaseQuantizerConfig):
    """dtype (None disables), zero-point domain, group shapes per scale level and the scale dtype per level."""

    dtype: QuantDataType | None = None
    zero_point: ZeroPointDomain | None = None
    group_shapes: tp.Sequence[tp.Sequence[int]] = ((-1, -1, -1),)
    scale_dtypes: tp.Sequence[torch.dtype | QuantDataType | None] = (None,)

    def __post_init__(self) -> None:
        self.group_shapes, self.scale_dtypes = ShapeUtils.format_group_configs(group_shapes=self.group_shapes,
                                                                              scale_dtypes=self.scale_dtypes)
        if self.dtype is None:
            self.group_shapes, self.scale_dtypes = ((-1, -1, -1),), (None,)

    quant_dtype = property(lambda self: self.dtype)
    zero_domain = property(lambda self: self.zero_point)
    largest_group_shape = property(lambda self: self.group_shapes[0])
    smallest_group_shape = property(lambda self: self.group_shapes[-1])

    def decompose(self) -> DecomposedQuantizerConfig:
        return DecomposedQuantizerConfig(steps=(self,) if self.dtype is not None else ())


@configclass
@dataclass
class ProgressiveQuantizerConfig(QuantizerConfig):
    """Multi-step quantization: the tensor is first quantized to ``intermediate_dtypes[i]`` using the scale
    levels up to ``intermediate_levels[i]``, and that result is quantized further with the remaining levels."""

    intermediate_dtypes: tp.Sequence[QuantDataType] = field(default_factory=tuple)
    intermediate_levels: tp.Sequence[int] = field(default_factory=tuple)
    needs_dequant_saturation: bool = False

    def __post_init__(self) -> None:
        super().__post_init__()
        if self.dtype is None:
            self.intermediate_dtypes, self.intermediate_levels, self.needs_dequant_saturation = (), (), False
            return
        n_levels = len(self.group_shapes)
        if isinstance(self.intermediate_dtypes, QuantDataType):
            self.intermediate_dtypes = (self.intermediate_dtypes,)
        if isinstance(self.intermediate_levels, int):
            self.intermediate_levels = (self.intermediate_levels,)
        self.intermediate_dtypes = tuple(self.intermediate_dtypes)
        self.intermediate_levels = tuple(lv % n_levels for lv in self.intermediate_levels)
        if not self.intermediate_dtypes:
            self.intermediate_levels, self.needs_dequant_saturation = (), False
        assert len(self.intermediate_dtypes) == len(self.intermediate_levels) < n_levels
        assert all(isinstance(d, QuantDataType) for d in self.intermediate_dtypes)
        assert all(lv < n_levels - 1 for lv in self.intermediate_levels)

    def decompose(self) -> DecomposedQuantizerConfig:
        if self.dtype is None:
            return DecomposedQuantizerConfig(steps=())
        if not self.intermediate_dtypes:
            return DecomposedQuantizerConfig(steps=(self,))
        # (called once per quantize(): the step configs are rebuilt only when a field changed)
        key = (self.dtype, self.zero_point, self.group_shapes, self.scale_dtypes, self.intermediate_dtypes,
               self.intermediate_levels, self.needs_dequant_saturation)
        memo = self.__dict__.get("_decomposed")
        if memo is not None and memo[0] == key:
            return memo[1]
        self.__dict__["_decomposed"] = (key, self._decompose_steps())
        return self.__dict__["_decomposed"][1]

    def _decompose_steps(self) -> DecomposedQuantizerConfig:
        steps, first = [], 0
        for level, dtype in zip(self.intermediate_levels, self.intermediate_dtypes, strict=True):
            steps.append(QuantizerConfig(dtype=dtype, zero_point=None, group_shapes=self.group_shapes[first : level + 1],
                                         scale_dtypes=self.scale_dtypes[first : level + 1]))
            first = level + 1
        steps.append(QuantizerConfig(dtype=self.dtype, zero_point=self.zero_point,
                                     group_shapes=self.group_shapes[first:], scale_dtypes=self.scale_dtypes[first:]))
        return DecomposedQuantizerConfig(steps=tuple(steps), needs_dequant_saturation=self.needs_dequant_saturation)
