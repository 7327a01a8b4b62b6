"""Channel-reorder settings (reference: deepcompressor/calib/config/reorder.py:19-140)."""

import enum
from dataclasses import dataclass, field

from ...utils.config import SkipBasedConfig, configclass
from .search import (SearchBasedCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                     SearchBasedCalibStrategy)

__all__ = ["ChannelOrderCalibConfig", "SkipBasedChannelOrderConfig"]


@configclass
@dataclass
class ChannelOrderCalibConfig(SearchBasedCalibConfig):
    """Fields as in the reference.  ``channel_metric``: the per-channel statistic channels are ranked by (of the
    inputs, of the weights, or the product of both); ``channel_index``: ranked channels fill the quantization groups
    one after the other (Sequential) or are dealt out over them (Transpose)."""

    class ChannelMetric(enum.Enum):
        InputsAbsMax = "xMax"
        InputsAbsMean = "xAvg"
        InputsRootMeanSquare = "xRms"
        WeightsAbsMax = "wMax"
        WeightsAbsMean = "wAvg"
        WeightsRootMeanSquare = "wRms"
        AbsMaxProduct = "pMax"
        AbsMeanProduct = "pAvg"
        RootMeanSquareProduct = "pRms"

    class ChannelIndex(enum.Enum):
        Sequential = "Seq"
        Transpose = "Trp"

    objective: SearchBasedCalibObjective = field(init=False, default=SearchBasedCalibObjective.OutputsError)
    granularity: SearchBasedCalibGranularity = field(init=False, default=SearchBasedCalibGranularity.Layer)
    element_batch_size: int = field(init=False, default=-1)
    element_size: int = field(init=False, default=-1)
    pre_reshape: bool = field(init=False, default=True)
    allow_x_quant: bool = True
    allow_w_quant: bool = True
    channel_metric: ChannelMetric = ChannelMetric.InputsAbsMax
    channel_index: ChannelIndex = ChannelIndex.Sequential
    dynamic: bool = False

    def __post_init__(self) -> None:
        if self.strategy != SearchBasedCalibStrategy.Manual:
            self.strategy = SearchBasedCalibStrategy.GridSearch
        super().__post_init__()

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names = super().generate_dirnames(**kwargs)
        if self.strategy == SearchBasedCalibStrategy.Manual:
            name = f"{self.channel_metric.name}.{self.channel_index.name}"
        else:
            name = "search"
        names.append(name + (".dynamic" if self.dynamic else ""))
        off = [tag for tag, allowed in (("x", self.allow_x_quant), ("w", self.allow_w_quant)) if not allowed]
        if off:
            names.append(f"disallow.[{'+'.join(off)}]")
        return [f"{prefix}.{n}" for n in names] if prefix else names


@configclass
@dataclass
class SkipBasedChannelOrderConfig(SkipBasedConfig, ChannelOrderCalibConfig):
    pass
