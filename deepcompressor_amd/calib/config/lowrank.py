"""Low-rank branch calibration settings (reference: deepcompressor/calib/config/lowrank.py:16-110)."""

from dataclasses import dataclass, field

from ...quantizer.config.lowrank import QuantLowRankConfig
from ...utils.common import num2str
from ...utils.config import SkipBasedConfig, configclass
from .search import SearchBasedCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibStrategy

__all__ = ["QuantLowRankCalibConfig", "SkipBasedQuantLowRankCalibConfig"]


@configclass
@dataclass
class QuantLowRankCalibConfig(SearchBasedCalibConfig, QuantLowRankConfig):
    """``num_iters`` alternations of "branch = SVD_r(W - Q)" and "Q = quantize(W - branch)"; ``early_stop`` ends
    them at the first iteration that does not improve the error."""

    granularity: SearchBasedCalibGranularity = field(init=False, default=SearchBasedCalibGranularity.Layer)
    element_batch_size: int = field(init=False, default=-1)
    element_size: int = field(init=False, default=-1)
    pre_reshape: bool = field(init=False, default=True)
    num_iters: int = 1
    early_stop: bool = False

    def __post_init__(self):
        if self.strategy != SearchBasedCalibStrategy.Manual:
            self.strategy = SearchBasedCalibStrategy.GridSearch
        if self.compensate and self.num_iters <= 1:
            self.exclusive = True
        super().__post_init__()

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names = SearchBasedCalibConfig.generate_dirnames(self, **kwargs)
        name = f"i{num2str(self.num_iters)}.r{num2str(self.rank)}"
        name += ".exclusive" if self.exclusive else ""
        name += ".compensate" if self.compensate else ""
        name += ".earlystop" if self.early_stop and self.num_iters > 1 else ""
        names.append(name)
        return [f"{prefix}.{n}" for n in names] if prefix else names


@configclass
@dataclass
class SkipBasedQuantLowRankCalibConfig(SkipBasedConfig, QuantLowRankCalibConfig):
    pass
