from .linear import ConcatLinear, ShiftedLinear
from .lowrank import AccumBranchHook, LowRankBranch
from .sdpa import ScaleDotProductAttention

__all__ = ["LowRankBranch", "AccumBranchHook", "ConcatLinear", "ShiftedLinear", "ScaleDotProductAttention"]
