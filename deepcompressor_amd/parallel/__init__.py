"""Multi-GPU partitioning of the calibration hot path.

Linears (and whole layers) are independent once their activations are given, so work is sharded across one
process per GPU with no collective in the data path; the only exchanges are a barrier and tiny reductions of
timings / per-tensor statistics (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""

from __future__ import annotations

import torch

__all__ = ["assign_round_robin", "assign_by_cost", "barrier_and_max", "all_reduce_sum"]


def assign_round_robin(num_items: int, world: int, rank: int) -> list[int]:
    """Indices of the items (layers) rank ``rank`` owns."""
    return list(range(rank, num_items, world))


def assign_by_cost(costs: list[float], world: int) -> list[list[int]]:
    """Longest-processing-time-first partition of items with the given costs; returns one index list per rank."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads, parts = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def _dist():
    import torch.distributed as dist

    return dist


def barrier_and_max(value: float, world: int, device: torch.device | str = "cpu") -> float:
    """Barrier, then the maximum of ``value`` over all ranks (identity for one rank)."""
    if world <= 1:
        return value
    dist = _dist()
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(t: torch.Tensor, world: int) -> torch.Tensor:
    """Sum of a small statistics tensor (per-tensor errors, per-channel spans) over ranks, in place."""
    if world > 1:
        _dist().all_reduce(t)
    return t
