from .lowrank import AccumBranchHook, LowRankBranch

__all__ = ["LowRankBranch", "AccumBranchHook"]
