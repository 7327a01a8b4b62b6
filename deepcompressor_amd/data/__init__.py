from .cache import IOTensorsCache, TensorCache, TensorsCache
from .codebook import Codebook
from .common import TensorType
from .dtype import QDType, QuantDataType
from .range import DynamicRange, LogQuantRange, ProtectiveQuantRange, QuantRange, RangeBound
from .scale import QuantScale
from .tensor import QuantTensor
from .zero import ZeroPointDomain

__all__ = ["IOTensorsCache", "TensorCache", "TensorsCache", "Codebook", "TensorType", "QDType", "QuantDataType",
           "DynamicRange", "LogQuantRange", "ProtectiveQuantRange", "QuantRange", "RangeBound", "QuantScale",
           "QuantTensor", "ZeroPointDomain"]
