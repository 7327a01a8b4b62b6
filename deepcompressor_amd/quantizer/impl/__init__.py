from .base import QuantizerImpl
from .info import QuantInfo, QuantScaleInfo, QuantStepInfo
from .simple import simple_quantize

__all__ = ["QuantizerImpl", "QuantInfo", "QuantScaleInfo", "QuantStepInfo", "simple_quantize"]
