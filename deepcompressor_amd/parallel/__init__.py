"""Multi-GPU partitioning of the calibration hot path.

Linears (and whole layers) are independent once their activations are given, so work is sharded across one
process per GPU with no collective in the data path; the only exchanges are a barrier and tiny reductions of
timings / per-tensor statistics (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""

from __future__ import annotations

import torch

__all__ = ["assign_round_robin", "assign_by_cost", "barrier_and_max", "all_reduce_sum", "row_shard", "map_rows_sharded"]


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


def row_shard(num_rows: int, world: int, rank: int, multiple: int = 1) -> tuple[int, int]:
    """[begin, end) of the output rows rank ``rank`` owns when ``num_rows`` rows are split into ``world`` contiguous
    slabs whose sizes are multiples of ``multiple`` (the row-group size, so no quantization group is cut)."""
    assert num_rows % multiple == 0, "rows must be a whole number of row groups"
    units = num_rows // multiple
    base, extra = divmod(units, world)
    begin = rank * base + min(rank, extra)
    return begin * multiple, (begin + base + (1 if rank < extra else 0)) * multiple


def map_rows_sharded(fn, weight: torch.Tensor, world: int, rank: int, multiple: int = 1) -> torch.Tensor:
    """Row-sharded evaluation of a row-independent operator inside ONE linear: fake quantization, the per-group range
    search and the GPTQ sweep (given the shared Hessian) touch only their own output rows, so each rank runs ``fn``
    on its slab ``weight[begin:end]`` and the slabs are all-gathered -- the only collective, of the result itself.
    Per-group results do not depend on the sharding."""
    if world <= 1:
        return fn(weight)
    dist = _dist()
    begin, end = row_shard(weight.shape[0], world, rank, multiple)
    mine = fn(weight[begin:end]).contiguous()
    sizes = [row_shard(weight.shape[0], world, r, multiple) for r in range(world)]
    parts = [torch.empty((e - b, *mine.shape[1:]), dtype=mine.dtype, device=mine.device) for b, e in sizes]
    dist.all_gather(parts, mine) if len({e - b for b, e in sizes}) == 1 else _all_gather_ragged(parts, mine, rank, world)
    return torch.cat(parts, dim=0)


def _all_gather_ragged(parts: list[torch.Tensor], mine: torch.Tensor, rank: int, world: int) -> None:
    dist = _dist()
    for r in range(world):
        if r == rank:
            parts[r].copy_(mine)
        dist.broadcast(parts[r], src=r)
