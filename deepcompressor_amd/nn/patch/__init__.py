from .conv import ConcatConv2d, ShiftedConv2d
from .linear import ConcatLinear, ShiftedLinear
from .lowrank import AccumBranchHook, LowRankBranch, add_low_rank_branches
from .sdpa import ScaleDotProductAttention

__all__ = ["LowRankBranch", "AccumBranchHook", "ConcatLinear", "ShiftedLinear", "ConcatConv2d", "ShiftedConv2d",
           "ScaleDotProductAttention"]
