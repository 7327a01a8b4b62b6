from .base import BaseQuantizerConfig, DecomposedQuantizerConfig, ProgressiveQuantizerConfig, QuantizerConfig
from .kernel import BaseKeyEnableQuantKernelConfig, BaseQuantKernel, BaseQuantKernelConfig
from .lowrank import QuantLowRankConfig

__all__ = ["BaseQuantizerConfig", "DecomposedQuantizerConfig", "ProgressiveQuantizerConfig", "QuantizerConfig",
           "BaseKeyEnableQuantKernelConfig", "BaseQuantKernel", "BaseQuantKernelConfig", "QuantLowRankConfig"]
