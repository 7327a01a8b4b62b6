"""Multi-GPU partitioning of the calibration hot path.

Linears (and whole layers) are independent once their activations are given, so work is sharded across one
process per GPU with no collective in the data path; the only exchanges are a barrier and tiny reductions of
timings / per-tensor statistics (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""

from __future__ import annotations

import torch

__all__ = ["assign_round_robin", "assign_by_cost", "barrier_and_max", "all_reduce_sum", "all_reduce_max", "row_shard", "map_rows_sharded", "all_gather_slabs",
           "search_range_row_sharded", "run_on_streams"]


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


def _collective_device(dist) -> torch.device:
    """Where a tensor must live for the default group's backend: RCCL ("nccl") only moves accelerator memory,
    gloo only host memory."""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier_and_max(value: float, world: int, device: torch.device | str | None = None) -> float:
    """Barrier, then the maximum of ``value`` over all ranks (identity for one rank).  ``device`` defaults to the
    one the backend can move (the current GPU under RCCL, the host under gloo)."""
    if world <= 1:
        return value
    dist = _dist()
    home = _collective_device(dist)
    dev = home if device is None else torch.device(device)
    if dev.type != home.type:
        dev = home
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _all_reduce(t: torch.Tensor, world: int, op) -> torch.Tensor:
    if world > 1:
        dist = _dist()
        home = _collective_device(dist)
        if t.device.type != home.type:  # GPU tensor on gloo, or host tensor on RCCL: make the trip, copy back
            moved = t.to(home)
            dist.all_reduce(moved, op=op)
            t.copy_(moved)
        else:
            dist.all_reduce(t, op=op)
    return t


def all_reduce_sum(t: torch.Tensor, world: int) -> torch.Tensor:
    """Sum of a small statistics tensor (per-tensor errors, partial Hessians) over ranks, in place.  With a backend
    that has no device path (gloo in the tests) a GPU tensor makes the trip through the host."""
    return _all_reduce(t, world, _dist().ReduceOp.SUM) if world > 1 else t


def all_reduce_max(t: torch.Tensor, world: int) -> torch.Tensor:
    """Element-wise maximum over ranks, in place (per-channel abs-max spans of token-sharded activations)."""
    return _all_reduce(t, world, _dist().ReduceOp.MAX) if world > 1 else t


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
    return all_gather_slabs(mine, [e - b for b, e in sizes], world, rank)


def all_gather_slabs(mine: torch.Tensor, rows: list[int], world: int, rank: int) -> torch.Tensor:
    """Concatenation along dim 0 of every rank's slab (rank r contributes ``rows[r]`` rows; ``mine`` is this rank's).
    The slabs travel in the memory the backend can move (device memory under RCCL, host memory under gloo) and the
    result comes back on ``mine``'s device."""
    if world <= 1:
        return mine
    dist = _dist()
    home = _collective_device(dist)
    sent = mine.contiguous() if mine.device.type == home.type else mine.to(home)
    parts = [torch.empty((n, *sent.shape[1:]), dtype=sent.dtype, device=sent.device) for n in rows]
    if len(set(rows)) == 1:
        dist.all_gather(parts, sent)
    else:
        _all_gather_ragged(parts, sent, rank, world)
    return torch.cat(parts, dim=0).to(mine.device)


def search_range_row_sharded(weight: torch.Tensor, config, quantizer, world: int, rank: int, multiple: int = 1):
    """Range search of ONE weight whose output rows are split over ``world`` processes (one per GPU).  Every rank
    searches the slab ``weight[begin:end]``; per-group (Group granularity) decisions need nothing from the others,
    while errors that are summed over rows (ChannelGroup: one per group of input channels, Layer: one per tensor)
    are all-reduced -- a table of candidates x channel groups float64 values per candidate batch, the only exchange
    -- before any rank compares them, so all ranks pick the same candidates.  Returns ``(begin, end, ranges)``, the
    per-step ranges of the rank's own rows."""
    import torch.nn as nn

    from ..calib.range import calibrate_dynamic_range
    from ..data.common import TensorType

    begin, end = row_shard(weight.shape[0], world, rank, multiple)
    slab = nn.Parameter(weight[begin:end].contiguous(), requires_grad=False)
    reduce = (lambda t: all_reduce_sum(t, world)) if world > 1 else None
    ranges = calibrate_dynamic_range(TensorType.Weights, config, True, quantizer, modules=[None], activations=None,
                                     weights=[slab], reduce_errors=reduce)
    return begin, end, ranges


def _all_gather_ragged(parts: list[torch.Tensor], mine: torch.Tensor, rank: int, world: int) -> None:
    dist = _dist()
    for r in range(world):
        if r == rank:
            parts[r].copy_(mine)
        dist.broadcast(parts[r], src=r)


_job_streams: dict = {}
_job_streams_lock = __import__("threading").Lock()


def _job_stream(device: torch.device, slot: int, priority: int) -> "torch.cuda.Stream":
    key = (device.index, slot, priority)
    with _job_streams_lock:
        stream = _job_streams.get(key)
        if stream is None:
            stream = _job_streams[key] = torch.cuda.Stream(device=device, priority=priority)
    return stream


def run_on_streams(jobs, device: torch.device | str | None = None, priorities: list[int] | None = None,
                   after: list | None = None) -> list:
    """Run independent jobs concurrently on ONE GPU, each on its own HIP stream driven by its own host thread, and
    return their results in order.

    The calibration pipeline of a linear alternates kernels that fill the chip (the Hessian SYRK, big GEMMs) with
    kernels that cannot (the 128-column sweep, the single-workgroup panel factorisation, small GEMMs) and with
    host-side checks that wait for the device (a failed factorisation must be seen to raise the damping).  Groups of
    linears that do not share activations are independent, so their pipelines may interleave: while one group's
    latency-bound kernels run, another group's matrix-core kernels use the idle CUs.  One thread per stream keeps
    each pipeline's host waits from stalling the others; per-thread state of the kernels (workspaces, memoised
    Hessians) keeps them apart.

    Every stream first waits for the work already queued on the caller's current stream (the inputs), and the
    caller's stream waits for every job's stream before this returns, so results can be used as usual afterwards.
    ``priorities``: optional HIP stream priority per job (lower number = higher priority).  ``after``: optional
    event per job, recorded on the caller's stream; job i's stream then waits for that event only, not for everything
    queued on the caller's stream -- a job whose inputs are ready early starts while the caller's stream is still
    producing the inputs of the others."""
    import threading

    jobs = list(jobs)
    if len(jobs) <= 1:
        return [job() for job in jobs]
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    caller = torch.cuda.current_stream(dev)
    # the same stream objects call after call: the caching allocator keeps a pool of freed blocks per stream, so a job
    # that lands on the stream its predecessor used finds its buffers there instead of allocating device memory anew
    streams = [_job_stream(dev, i, priorities[i] if priorities else 0) for i in range(len(jobs))]
    results, errors = [None] * len(jobs), [None] * len(jobs)
    grad_mode, inference = torch.is_grad_enabled(), torch.is_inference_mode_enabled()

    def work(i: int) -> None:
        try:
            torch.cuda.set_device(dev)
            if after is not None and after[i] is not None:
                streams[i].wait_event(after[i])
            else:
                streams[i].wait_stream(caller)
            with torch.inference_mode(inference), torch.set_grad_enabled(grad_mode), torch.cuda.stream(streams[i]):
                results[i] = jobs[i]()
        except BaseException as exc:  # noqa: BLE001 -- re-raised in the caller's thread
            errors[i] = exc

    threads = [threading.Thread(target=work, args=(i,), name=f"dcq-stream-{i}") for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in streams:
        caller.wait_stream(s)
    for exc in errors:
        if exc is not None:
            raise exc
    _record_stream(results, caller)
    return results


def _record_stream(obj, stream) -> None:
    """Tell the allocator that tensors made on a job's stream are now used on ``stream`` (so their memory is not
    handed out again on the job's stream while ``stream`` still reads it)."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)
