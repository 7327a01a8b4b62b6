from .lowrank import QuantLowRankCalibConfig, SkipBasedQuantLowRankCalibConfig
from .range import DynamicRangeCalibConfig, SkipBasedDynamicRangeCalibConfig
from .smooth import SkipBasedSmoothCalibConfig, SmoothAttentionCalibConfig, SmoothCalibConfig, SmoothSpanMode
from .search import (SearchBasedCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                     SearchBasedCalibStrategy)

__all__ = ["QuantLowRankCalibConfig", "SkipBasedQuantLowRankCalibConfig", "DynamicRangeCalibConfig", "SkipBasedDynamicRangeCalibConfig", "SearchBasedCalibConfig",
           "SearchBasedCalibGranularity", "SearchBasedCalibObjective", "SearchBasedCalibStrategy",
           "SkipBasedSmoothCalibConfig", "SmoothAttentionCalibConfig", "SmoothCalibConfig", "SmoothSpanMode"]
