"""Smooth-quantization settings (reference: deepcompressor/calib/config/smooth.py:28-260)."""

import enum
from dataclasses import dataclass, field

from ...utils.common import num2str
from ...utils.config import SkipBasedConfig, configclass
from .search import (SearchBasedCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                     SearchBasedCalibStrategy)

__all__ = ["SmoothSpanMode", "SmoothCalibConfig", "SmoothAttentionCalibConfig", "SkipBasedSmoothCalibConfig",
           "SmoothTransfomerConfig"]


class SmoothSpanMode(enum.Enum):
    """Which per-channel statistic feeds the smoothing scale."""

    AbsMax = enum.auto()
    RootMeanSquare = enum.auto()


def _beta_class(beta: float) -> str:
    return "pos" if beta > 0 else {0: "0", -1: "-1", -2: "-2"}.get(beta, "other")


# Candidate families of the grid search, per (alpha setting, beta setting).  "A0" = (a, 0), "0B" = (0, b),
# "A1" = (a, 1 - a), "AA" = (a, a), "AB" = every (a, b); a, b run over i / num_grids, i = 1 .. num_grids - 1.
_GRID_FAMILIES = {
    "pos": {"pos": ("AA",), "0": ("A0",), "-1": ("A1",), "-2": ("A0", "A1"), "other": ("A0", "AB")},
    "0": {"pos": ("0B",), "0": ("A0", "0B"), "-1": ("0B", "A1"), "-2": ("A0", "0B", "A1"), "other": ("A0", "0B", "AB")},
    "-1": {"pos": ("A1",), "-1": ("A1",), "0": ("A0", "A1"), "-2": ("A0", "A1"), "other": ("A0", "AB")},
    "-2": {"pos": ("0B", "A1"), "-1": ("0B", "A1"), "0": ("A0", "0B", "A1"), "-2": ("A0", "0B", "A1"),
           "other": ("A0", "0B", "AB")},
    "-3": {"pos": ("0B", "AB"), "0": ("0B", "A0", "AB"), "-1": ("0B", "A0", "AB"), "-2": ("0B", "A0", "AB"),
           "other": ("0B", "A0", "AB")},
}


@configclass
@dataclass
class SmoothCalibConfig(SearchBasedCalibConfig):
    """Fields as in the reference: ``spans`` pairs (statistic of the alpha tensor, statistic of the beta tensor);
    ``alpha`` / ``beta`` in [0, 1] are fixed exponents, the negative codes -1, -2, -3 select search families."""

    fuse_when_possible: bool = True
    allow_a_quant: bool = True
    allow_b_quant: bool = True
    spans: list = field(default_factory=list)
    a_spans: list = field(default_factory=list, init=False)
    b_spans: list = field(default_factory=list, init=False)
    alpha: float = 0.5
    beta: float = -1
    num_grids: int = 20
    allow_low_rank: bool = False

    def __post_init__(self) -> None:
        spans, self.a_spans, self.b_spans = [], [], []
        for a_span, b_span in self.spans:
            a_span = SmoothSpanMode[a_span] if isinstance(a_span, str) else a_span
            b_span = SmoothSpanMode[b_span] if isinstance(b_span, str) else b_span
            assert isinstance(a_span, SmoothSpanMode), f"Invalid span mode used for alpha: {a_span}"
            assert isinstance(b_span, SmoothSpanMode), f"Invalid span mode used for beta: {b_span}"
            if (a_span, b_span) in spans:
                continue
            spans.append((a_span, b_span))
            if a_span not in self.a_spans:
                self.a_spans.append(a_span)
            if b_span not in self.b_spans:
                self.b_spans.append(b_span)
        self.spans = spans
        if self.strategy == SearchBasedCalibStrategy.Manual:
            assert len(self.spans) == 1, "Only one span combination is allowed in manual mode"
            assert self.alpha != 0 or self.beta != 0, "alpha and beta cannot be both zero"
            self.alpha, self.beta = self.get_alpha_beta_pairs()[0]
        if self.granularity == SearchBasedCalibGranularity.Group:
            self.granularity = SearchBasedCalibGranularity.ChannelGroup
        if self.allow_low_rank:
            self.granularity = SearchBasedCalibGranularity.Layer
        assert -3 <= self.alpha <= 1, "alpha must be less than or equal to 1"
        assert -3 <= self.beta <= 1, "beta must be less than or equal to 1"
        super().__post_init__()

    def get_alpha_beta_pairs(self) -> list[tuple[float, float]]:
        """Manual: the single (alpha, beta), a negative entry meaning "one minus the other".  Grid search:
        (0, 0) followed by the families selected by the two settings."""
        if self.strategy == SearchBasedCalibStrategy.Manual:
            if self.beta < 0:
                assert 0 <= self.alpha <= 1, "alpha must be in [0, 1]"
                return [(self.alpha, 1 - self.alpha)]
            if self.alpha < 0:
                assert 0 <= self.beta <= 1, "beta must be in [0, 1]"
                return [(1 - self.beta, self.beta)]
            assert 0 <= self.alpha <= 1, "alpha must be in [0, 1]"
            assert 0 <= self.beta <= 1, "beta must be in [0, 1]"
            return [(self.alpha, self.beta)]
        grid = [i / self.num_grids for i in range(1, self.num_grids)]
        alpha_class = "pos" if self.alpha > 0 else {0: "0", -1: "-1", -2: "-2", -3: "-3"}.get(self.alpha)
        if alpha_class is None:
            raise ValueError("Invalid alpha and beta values")
        family = {"A0": [(a, 0) for a in grid], "0B": [(0, b) for b in grid], "A1": [(a, 1 - a) for a in grid],
                  "AA": [(a, a) for a in grid], "AB": [(a, b) for a in grid for b in grid]}
        pairs = [(0, 0)]
        for name in _GRID_FAMILIES[alpha_class][_beta_class(self.beta)]:
            pairs += family[name]
        return pairs

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        names = super().generate_dirnames(**kwargs)
        names.append("[{}]".format("+".join(f"a.{a.name}.b.{b.name}" for a, b in self.spans)))
        alpha, beta = num2str(self.alpha), num2str(self.beta)
        if self.strategy == SearchBasedCalibStrategy.Manual:
            names.append(f"a{alpha}.b{beta}")
        elif self.alpha > 0:
            names.append(f"g{self.num_grids}.b{beta}")
        elif self.beta > 0:
            names.append(f"g{self.num_grids}.a{alpha}")
        else:
            names.append(f"g{self.num_grids}.a{alpha}.b{beta}")
        if self.allow_low_rank:
            names[-1] += ".lr"
        if not self.fuse_when_possible:
            names[-1] += ".nf"
        disallows = [n for n, allowed in (("a", self.allow_a_quant), ("b", self.allow_b_quant)) if not allowed]
        if disallows:
            names.append(f"disallow.[{'+'.join(disallows)}]")
        return [f"{prefix}.{name}" for name in names] if prefix else names


@configclass
@dataclass
class SkipBasedSmoothCalibConfig(SkipBasedConfig, SmoothCalibConfig):
    pass


@configclass
@dataclass
class SmoothAttentionCalibConfig(SmoothCalibConfig):
    """Attention (query / key) smoothing: the search always scores the outputs of the whole layer."""

    objective: SearchBasedCalibObjective = field(init=False, default=SearchBasedCalibObjective.OutputsError)
    granularity: SearchBasedCalibGranularity = field(init=False, default=SearchBasedCalibGranularity.Layer)
    element_batch_size: int = field(init=False, default=-1)
    element_size: int = field(init=False, default=-1)
    pre_reshape: bool = field(init=False, default=True)
    allow_low_rank: bool = field(init=False, default=False)


@configclass
@dataclass
class SmoothTransfomerConfig:
    """Smoothing of a transformer block (calib/config/smooth.py:349-400): ``proj`` for the linears fed by one
    activation, ``attn`` for the query-key product."""

    proj: SkipBasedSmoothCalibConfig | None = None
    attn: SmoothAttentionCalibConfig | None = None

    @property
    def enabled_proj(self) -> bool:
        return self.proj is not None

    @property
    def enabled_attn(self) -> bool:
        return self.attn is not None

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        proj_names = self.proj.generate_dirnames(prefix="proj") if self.proj is not None else []
        attn_names = self.attn.generate_dirnames(prefix="attn") if self.attn is not None else []
        names = []
        for index in range(max(len(proj_names), len(attn_names))):
            name = []
            if index < len(proj_names):
                name.append(proj_names[index])
            if index < len(attn_names):
                name.append(attn_names[index])
            names.append("-".join(name))
        return [f"{prefix}.{name}" for name in names] if prefix else names
