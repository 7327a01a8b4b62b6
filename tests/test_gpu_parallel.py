"""Range search of one weight with its rows sharded over two processes (both on the one GPU of the test box, gloo for
the exchange): errors that are summed over rows are all-reduced before they are compared, so the shards decide like
the unsharded search.  On the 8-GPU node the same code runs one process per GPU with RCCL."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CASES = [  # (granularity, group size, rows): 128 -> the fused search kernel, 96 -> the candidate-by-candidate path
    ("ChannelGroup", 128, 192), ("Layer", 128, 192), ("ChannelGroup", 96, 64), ("Layer", 96, 64), ("Group", 128, 192)]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _search(world: int, rank: int, granularity: str, group: int, rows: int):
    from deepcompressor_amd.calib.config import (DynamicRangeCalibConfig, SearchBasedCalibGranularity,
                                                 SearchBasedCalibObjective, SearchBasedCalibStrategy)
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.parallel import search_range_row_sharded
    from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

    g = torch.Generator().manual_seed(7)
    w = torch.randn(rows, 768, generator=g) * 0.02
    w = torch.where(torch.rand(rows, 768, generator=g) < 0.01, w * 6, w).to(torch.bfloat16).cuda()
    q = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, group),),
                                  scale_dtypes=(torch.float16,)), develop_dtype=torch.float32)
    cfg = DynamicRangeCalibConfig(degree=2.4, objective=SearchBasedCalibObjective.TensorError,
                                  strategy=SearchBasedCalibStrategy.GridSearch,
                                  granularity=SearchBasedCalibGranularity[granularity], max_shrink=0.2, max_expand=1.0,
                                  num_grids=20)
    begin, end, (rng,) = search_range_row_sharded(w, cfg, q, world, rank, multiple=1)
    return begin, end, rng.max.float().cpu().numpy()


def _worker(rank: int, world: int, port: int, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        out.put((rank, [_search(world, rank, *case) for case in CASES]))
    finally:
        dist.destroy_process_group()


def test_row_sharded_search_decides_like_the_unsharded_one():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, case in enumerate(CASES):
        _, _, whole = _search(1, 0, *case)
        parts = [results[r][i] for r in range(world)]
        assert [(b, e) for b, e, _ in parts] == [(0, case[2] // 2), (case[2] // 2, case[2])]
        np.testing.assert_array_equal(np.concatenate([m for _, _, m in parts], axis=0), whole, err_msg=str(case))
    # the search did shrink some ranges (otherwise equal results would prove nothing)
    _, _, plain = _search(1, 0, "Group", 128, 192)
    g = torch.Generator().manual_seed(7)
    w = torch.randn(192, 768, generator=g) * 0.02
    w = torch.where(torch.rand(192, 768, generator=g) < 0.01, w * 6, w).to(torch.bfloat16).float()
    assert (plain.reshape(192, 6) < w.view(192, 6, 128).abs().amax(dim=2).numpy()).mean() > 0.2
