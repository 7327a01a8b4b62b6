from .config import (DynamicRangeCalibConfig, SearchBasedCalibConfig, SearchBasedCalibGranularity,
                     SearchBasedCalibObjective, SearchBasedCalibStrategy, SkipBasedDynamicRangeCalibConfig)
from .range import DynamicRangeCalibrator, calibrate_dynamic_range

__all__ = ["DynamicRangeCalibConfig", "SearchBasedCalibConfig", "SearchBasedCalibGranularity",
           "SearchBasedCalibObjective", "SearchBasedCalibStrategy", "SkipBasedDynamicRangeCalibConfig",
           "DynamicRangeCalibrator", "calibrate_dynamic_range"]
