"""Dynamic-range calibration settings (reference: deepcompressor/calib/config/range.py:15-153)."""

from dataclasses import dataclass

from ...utils.common import num2str
from ...utils.config import SkipBasedConfig, configclass
from .search import SearchBasedCalibConfig, SearchBasedCalibStrategy

__all__ = ["DynamicRangeCalibConfig", "SkipBasedDynamicRangeCalibConfig"]


@configclass
@dataclass
class DynamicRangeCalibConfig(SearchBasedCalibConfig):
    ratio: float = 1.0
    max_shrink: float = 0.2
    max_expand: float = 1.0
    num_grids: int = 80
    allow_scale: bool = False

    def get_linear_ratios(self) -> list[float]:
        """Shrink grid 1 - k/N (1 - max_shrink), then (if max_expand > 1) the expand grid, as python floats."""
        n, lo, hi = self.num_grids, self.max_shrink, self.max_expand
        assert lo < 1, "maximal shrinkage ratio must be less than 1"
        ratios = [1 - k / n * (1 - lo) for k in range(1, n + 1)]
        if hi > 1:
            ratios += [1 + k / n * (hi - 1) for k in range(1, n + 1)]
        return ratios

    def get_ratios(self) -> list[list[float]]:
        if self.strategy == SearchBasedCalibStrategy.Manual:
            return [[self.ratio]]
        if self.strategy == SearchBasedCalibStrategy.GridSearch:
            return [[1.0], self.get_linear_ratios()]
        raise ValueError(f"Invalid strategy: {self.strategy}")

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names = super().generate_dirnames(**kwargs)
        if self.strategy == SearchBasedCalibStrategy.Manual:
            name = f"r.[{num2str(self.ratio)}]"
        elif self.strategy == SearchBasedCalibStrategy.GridSearch:
            name = f"r.[{num2str(self.max_shrink)}.{num2str(self.max_expand)}].g{self.num_grids}"
        else:
            raise ValueError(f"Invalid strategy: {self.strategy}")
        names.append(name + (".scale" if self.allow_scale else ""))
        return [f"{prefix}.{n}" for n in names] if prefix else names


@configclass
@dataclass
class SkipBasedDynamicRangeCalibConfig(SkipBasedConfig, DynamicRangeCalibConfig):
    pass
