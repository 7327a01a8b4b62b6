"""The multi-process partitioning helpers under torch.distributed (gloo, world_size 2, CPU)."""

import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepcompressor_amd.parallel import all_reduce_sum, assign_by_cost, assign_round_robin, barrier_and_max


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
        out.put((rank, layers, elapsed, stats.tolist()))
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
    all_layers = sorted(i for _, layers, _, _ in results for i in layers)
    assert all_layers == list(range(32)), "every layer is owned by exactly one rank"
    assert all(elapsed == 2.0 for _, _, elapsed, _ in results), "max over ranks"
    assert all(stats == [32.0, 1.0] for _, _, _, stats in results)


def test_cost_balanced_assignment():
    # Llama-3-8B linears of one layer, cost ~ Hessian flops (C^2) + sweep (R*C): down_proj dominates
    costs = [4096**2, 4096**2 / 4, 4096**2 / 4, 4096**2, 14336 * 4096, 14336 * 4096, 14336**2]
    parts = assign_by_cost(costs, 2)
    assert sorted(i for p in parts for i in p) == list(range(7))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) == 14336**2  # the largest item alone on one rank is optimal here
    assert assign_by_cost([1.0] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert barrier_and_max(3.5, 1) == 3.5
