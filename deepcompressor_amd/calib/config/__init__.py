from .range import DynamicRangeCalibConfig, SkipBasedDynamicRangeCalibConfig
from .search import (SearchBasedCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                     SearchBasedCalibStrategy)

__all__ = ["DynamicRangeCalibConfig", "SkipBasedDynamicRangeCalibConfig", "SearchBasedCalibConfig",
           "SearchBasedCalibGranularity", "SearchBasedCalibObjective", "SearchBasedCalibStrategy"]
