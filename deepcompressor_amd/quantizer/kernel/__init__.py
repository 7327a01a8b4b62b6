from .rtn import QuantRtnKernel, rtn_quantize

__all__ = ["QuantRtnKernel", "rtn_quantize"]
