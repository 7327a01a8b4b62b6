"""Search-based calibration settings (reference: deepcompressor/calib/config/search.py:19-126)."""

import enum
from dataclasses import dataclass

from ...utils.common import num2str
from ...utils.config import configclass

__all__ = ["SearchBasedCalibStrategy", "SearchBasedCalibGranularity", "SearchBasedCalibObjective",
           "SearchBasedCalibConfig"]


class SearchBasedCalibStrategy(enum.Enum):
    Manual = enum.auto()
    GridSearch = enum.auto()


class SearchBasedCalibGranularity(enum.Enum):
    Group = enum.auto()
    ChannelGroup = enum.auto()
    Layer = enum.auto()


class SearchBasedCalibObjective(enum.Enum):
    TensorError = enum.auto()  # error of the quantized tensor itself
    ProductsError = enum.auto()  # error of the partial products x @ w
    OutputsError = enum.auto()  # error of the evaluation module's outputs


@configclass
@dataclass
class SearchBasedCalibConfig:
    degree: int = 2
    objective: SearchBasedCalibObjective = SearchBasedCalibObjective.OutputsError
    strategy: SearchBasedCalibStrategy = SearchBasedCalibStrategy.Manual
    granularity: SearchBasedCalibGranularity = SearchBasedCalibGranularity.Layer
    element_batch_size: int = -1
    sample_batch_size: int = -1
    element_size: int = -1
    sample_size: int = -1
    pre_reshape: bool = True
    outputs_device: str = "cpu"

    def __post_init__(self) -> None:
        if self.outputs_device != "cpu":
            self.outputs_device = None
        if self.element_size != 0 or self.sample_size != 0:
            assert self.element_batch_size != 0, "element_batch_size must not be zero"
            assert self.sample_batch_size != 0, "sample_batch_size must not be zero"
            assert self.element_size != 0, "element_size must not be zero"
            assert self.sample_size != 0, "sample_size must not be zero"
        else:
            assert self.objective == SearchBasedCalibObjective.TensorError
        if (self.objective != SearchBasedCalibObjective.TensorError
                and self.granularity == SearchBasedCalibGranularity.Layer):
            self.objective = SearchBasedCalibObjective.OutputsError
            self.element_batch_size = self.element_size = -1

    @property
    def needs_search(self) -> bool:
        return self.strategy != SearchBasedCalibStrategy.Manual

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        name = (f"{self.objective.name}.{self.strategy.name}.{self.granularity.name}.d{num2str(self.degree)}"
                f".e{num2str(self.element_size)}.s{num2str(self.sample_size)}")
        return [f"{prefix}.{name}" if prefix else name]
