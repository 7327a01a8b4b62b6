"""Quantization-kernel slot (reference: deepcompressor/quantizer/config/kernel.py:18-164).

A kernel object turns the float32 working tensor plus its per-group scale / zero into integer (or
codebook) values of shape ``view_shape``; ``Quantizer.quantize(kernel=...)`` accepts any object with
this interface.  The kernels of this package (round-to-nearest, GPTQ) run as HIP code.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass, field, fields

import torch

from ...data.dtype import QuantDataType
from ...data.range import QuantRange
from ...data.zero import ZeroPointDomain
from ...utils.config import EnableConfig, IncludeBasedConfig, KeyEnableConfig, configclass

__all__ = ["BaseQuantKernel", "BaseQuantKernelConfig", "BaseKeyEnableQuantKernelConfig"]


class BaseQuantKernel(ABC):
    @abstractmethod
    def quantize(self, tensor: torch.Tensor, *, view_shape: torch.Size, quant_dtype: QuantDataType,
                 zero_domain: ZeroPointDomain | None, scale: torch.Tensor, zero: torch.Tensor,
                 quant_range: QuantRange | None = None, **kwargs) -> torch.Tensor:
        """Return the quantized values (float32) in the shape ``view_shape``."""
        ...


class BaseQuantKernelConfig(ABC):
    @property
    @abstractmethod
    def name(self) -> str: ...

    @abstractmethod
    def build(self) -> BaseQuantKernel: ...

    @abstractmethod
    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]: ...


@configclass
@dataclass
class BaseKeyEnableQuantKernelConfig(KeyEnableConfig, EnableConfig):
    """Container whose fields are include-based kernel configs; picks the kernel for a module key."""

    _names: list[str] = field(init=False, repr=False, compare=False, default_factory=list)
    _kernels: dict[str, BaseQuantKernelConfig | None] = field(init=False, repr=False, compare=False,
                                                             default_factory=dict)

    def __post_init__(self) -> None:
        self.organize()

    def is_enabled(self) -> bool:
        return bool(self._kernels)

    def is_enabled_for(self, key: str) -> bool:
        return key in self._kernels

    def specialize_for(self, key: str) -> BaseQuantKernelConfig | None:
        return self._kernels.get(key, None)

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names: list[str] = []
        if self.is_enabled():
            for name in self._names:
                cfg = getattr(self, name)
                if cfg is not None and cfg.is_enabled():
                    names.extend(cfg.generate_dirnames(prefix=prefix, **kwargs))
        return names

    def organize(self) -> None:
        self._kernels.clear()
        self._names.clear()
        for f in fields(self):
            if f.name.startswith("_"):
                continue
            self._names.append(f.name)
            cfg = getattr(self, f.name)
            if cfg is None:
                continue
            assert isinstance(cfg, IncludeBasedConfig) and isinstance(cfg, BaseQuantKernelConfig), (
                f"Field '{f.name}' must be an include-based kernel config")
            if not cfg.is_enabled():
                setattr(self, f.name, None)
                continue
            for key in cfg.includes:
                assert key not in self._kernels, f"Key '{key}' is already included in other kernel configurations."
                self._kernels[key] = cfg
