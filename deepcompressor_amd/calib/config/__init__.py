from .lowrank import QuantLowRankCalibConfig, SkipBasedQuantLowRankCalibConfig
from .reorder import ChannelOrderCalibConfig, SkipBasedChannelOrderConfig
from .range import DynamicRangeCalibConfig, SkipBasedDynamicRangeCalibConfig
from .rotation import QuantRotationConfig
from .smooth import (SkipBasedSmoothCalibConfig, SmoothAttentionCalibConfig, SmoothCalibConfig, SmoothSpanMode,
                     SmoothTransfomerConfig)
from .search import (SearchBasedCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                     SearchBasedCalibStrategy)

__all__ = ["QuantRotationConfig", "SmoothTransfomerConfig", "ChannelOrderCalibConfig", "SkipBasedChannelOrderConfig", "QuantLowRankCalibConfig", "SkipBasedQuantLowRankCalibConfig", "DynamicRangeCalibConfig", "SkipBasedDynamicRangeCalibConfig", "SearchBasedCalibConfig",
           "SearchBasedCalibGranularity", "SearchBasedCalibObjective", "SearchBasedCalibStrategy",
           "SkipBasedSmoothCalibConfig", "SmoothAttentionCalibConfig", "SmoothCalibConfig", "SmoothSpanMode"]
