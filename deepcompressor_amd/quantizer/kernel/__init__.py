from .gptq import QuantGptqConfig, QuantGptqKernel, gptq_quantize
from .rtn import QuantRtnKernel, rtn_quantize

__all__ = ["QuantGptqConfig", "QuantGptqKernel", "gptq_quantize", "QuantRtnKernel", "rtn_quantize"]
