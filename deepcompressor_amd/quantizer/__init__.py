from .config import (BaseKeyEnableQuantKernelConfig, BaseQuantizerConfig, BaseQuantKernel, BaseQuantKernelConfig,
                     DecomposedQuantizerConfig, ProgressiveQuantizerConfig, QuantizerConfig, QuantLowRankConfig)
from .kernel import QuantGptqConfig, QuantGptqKernel, QuantRtnKernel
from .processor import Quantizer

__all__ = ["Quantizer", "QuantizerConfig", "ProgressiveQuantizerConfig", "DecomposedQuantizerConfig",
           "BaseQuantizerConfig", "BaseQuantKernel", "BaseQuantKernelConfig", "BaseKeyEnableQuantKernelConfig",
           "QuantLowRankConfig", "QuantRtnKernel", "QuantGptqConfig", "QuantGptqKernel"]
