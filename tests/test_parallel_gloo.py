"""The multi-process partitioning helpers under torch.distributed (gloo, world_size 2, CPU)."""

import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepcompressor_amd.parallel import (all_reduce_max, all_reduce_sum, assign_by_cost, assign_round_robin, barrier_and_max,
                                         map_rows_sharded, row_shard)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        layers = assign_round_robin(32, world, rank)
        # every rank times its own shard; the job time is the slowest rank's
        elapsed = barrier_and_max(1.0 + rank, world)
        # per-tensor statistics (e.g. layer-granularity errors) are summed over ranks
        stats = all_reduce_sum(torch.tensor([float(len(layers)), float(rank)]), world)
        # per-channel abs-max spans of token-sharded activations take the maximum over ranks
        spans = all_reduce_max(torch.tensor([1.0 + rank, 5.0 - rank]), world)
        assert spans.tolist() == [float(world), 5.0]
        # rows of one linear sharded over ranks: a row-independent operator (stand-in for quantize) gives the same
        # result as unsharded, for even and for ragged splits
        g = torch.Generator().manual_seed(0)
        rowwise = lambda w: torch.round(w / w.abs().amax(dim=1, keepdim=True) * 7)  # noqa: E731
        ok = True
        for rows, multiple in ((64, 1), (70, 2), (12, 4)):
            w = torch.randn(rows, 32, generator=g)
            ok = ok and torch.equal(map_rows_sharded(rowwise, w, world, rank, multiple), rowwise(w))
        out.put((rank, layers, elapsed, stats.tolist(), ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_reductions():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_layers = sorted(i for _, layers, _, _, _ in results for i in layers)
    assert all_layers == list(range(32)), "every layer is owned by exactly one rank"
    assert all(elapsed == 2.0 for _, _, elapsed, _, _ in results), "max over ranks"
    assert all(stats == [32.0, 1.0] for _, _, _, stats, _ in results)
    assert all(ok for *_, ok in results), "row-sharded results equal the unsharded ones"


def test_row_shard_partition():
    for rows, world, multiple in ((4096, 8, 1), (14336, 8, 128), (70, 4, 2), (12, 8, 4), (5, 2, 1)):
        spans = [row_shard(rows, world, r, multiple) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "contiguous, no overlap"
        assert all((e - b) % multiple == 0 for b, e in spans)
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= multiple


def test_cost_balanced_assignment():
    # Llama-3-8B linears of one layer, cost ~ Hessian flops (C^2) + sweep (R*C): down_proj dominates
    costs = [4096**2, 4096**2 / 4, 4096**2 / 4, 4096**2, 14336 * 4096, 14336 * 4096, 14336**2]
    parts = assign_by_cost(costs, 2)
    assert sorted(i for p in parts for i in p) == list(range(7))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) == 14336**2  # the largest item alone on one rank is optimal here
    assert assign_by_cost([1.0] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert barrier_and_max(3.5, 1) == 3.5
