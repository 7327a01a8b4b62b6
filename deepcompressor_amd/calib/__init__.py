from .config import (DynamicRangeCalibConfig, SearchBasedCalibConfig, SearchBasedCalibGranularity,
                     SearchBasedCalibObjective, SearchBasedCalibStrategy, SkipBasedDynamicRangeCalibConfig)
from .config.reorder import ChannelOrderCalibConfig, SkipBasedChannelOrderConfig
from .range import DynamicRangeCalibrator, calibrate_dynamic_range
from .reorder import ChannelOrderCalibrator, ChannelReorderer
from .rotate import (get_rotation_matrix, hadamard_in_channels, rotate_in_channels, rotate_out_channels,
                     transform_layer_norm_to_rms_norm, transform_norm_and_linear, transform_rms_norm_and_linear)

__all__ = ["ChannelOrderCalibConfig", "SkipBasedChannelOrderConfig", "ChannelOrderCalibrator", "ChannelReorderer",
           "DynamicRangeCalibConfig", "SearchBasedCalibConfig", "SearchBasedCalibGranularity",
           "SearchBasedCalibObjective", "SearchBasedCalibStrategy", "SkipBasedDynamicRangeCalibConfig",
           "DynamicRangeCalibrator", "calibrate_dynamic_range", "get_rotation_matrix", "hadamard_in_channels",
           "rotate_in_channels", "rotate_out_channels", "transform_layer_norm_to_rms_norm", "transform_norm_and_linear",
           "transform_rms_norm_and_linear"]
