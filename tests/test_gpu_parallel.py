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


def _gptq_worker(rank: int, world: int, port: int, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        out.put((rank, _gptq(world, rank)))
    finally:
        dist.destroy_process_group()


def _gptq(world: int, rank: int):
    """GPTQ of one 64 x 256 linear on 4 batches of 128 tokens; with two ranks each holds two of the batches."""
    from deepcompressor_amd.data.cache import TensorCache
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer import QuantGptqConfig, Quantizer, QuantizerConfig
    from deepcompressor_amd.quantizer.kernel import gptq as gm

    g = torch.Generator().manual_seed(21)
    w = (torch.randn(64, 256, generator=g) * 0.03).to(torch.bfloat16).cuda()
    gains = torch.exp(torch.randn(256, generator=g) * 0.7)
    xs = [(torch.randn(1, 128, 256, generator=g) * gains).to(torch.bfloat16).cuda() for _ in range(4)]
    mine = xs[rank::world]
    cache = TensorCache(mine, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=len(mine), num_total=len(mine),
                        num_samples=len(mine), orig_device=torch.device("cuda"))
    (hessian,) = gm.prepare_hessians([cache], torch.device("cuda", 0), world=world)
    q = Quantizer(QuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 64),),
                                  scale_dtypes=(torch.float16,)), kernel=QuantGptqConfig(), develop_dtype=torch.float32)
    out = q.quantize(w, inputs=cache).data
    gm.clear_hessian_cache()
    return hessian.cpu().numpy(), out.float().cpu().numpy()


def test_token_sharded_hessian_gives_the_unsharded_gptq_result():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gptq_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    h_whole, w_whole = _gptq(1, 0)
    (h0, w0), (h1, w1) = results[0], results[1]
    np.testing.assert_array_equal(h0, h1)  # one all-reduce: both ranks hold the same Hessian
    np.testing.assert_array_equal(w0, w1)
    assert np.abs(h0 - h_whole).max() <= 1e-5 * np.abs(h_whole).max()  # float32 sums in another order
    assert (w0 == w_whole).mean() >= 0.99


class _Pair(torch.nn.Module):
    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b

    def forward(self, x):
        return torch.cat([self.a(x), self.b(x)], dim=-1)


def _smooth(world: int, rank: int, strategy: str):
    """Smoothing scale of two linears that share their input, 4 batches of tokens; with two ranks two batches each."""
    import torch.nn as nn

    from deepcompressor_amd.calib.config import (SearchBasedCalibStrategy, SmoothCalibConfig, SmoothSpanMode)
    from deepcompressor_amd.calib.smooth import smooth_linear_modules
    from deepcompressor_amd.data.cache import TensorCache, TensorsCache
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
    from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

    g = torch.Generator().manual_seed(33)
    a, b = nn.Linear(64, 40, bias=False, device="cuda"), nn.Linear(64, 24, bias=False, device="cuda")
    a.weight.data = (torch.randn(40, 64, generator=g) * 0.08).cuda()
    b.weight.data = (torch.randn(24, 64, generator=g) * 0.08).cuda()
    gains = torch.exp(torch.randn(64, generator=g) * 1.2)
    xs = [(torch.randn(2, 12, 64, generator=g) * gains).cuda() for _ in range(4)]
    mine = xs[rank::world]
    cache = TensorsCache(TensorCache(mine, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=len(mine),
                                     num_total=len(mine), num_samples=2 * len(mine), orig_device=torch.device("cuda")))
    wq = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 16),),
                                   scale_dtypes=(torch.float16,)), key="w")
    xq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),),
                                   scale_dtypes=(torch.float16,)), channels_dim=-1, key="x")
    manual = strategy == "Manual"
    cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.3 if manual else 0.5,
                            beta=0.7 if manual else -2, num_grids=10, strategy=SearchBasedCalibStrategy[strategy])
    scale = smooth_linear_modules(None, [a, b], scale=None, config=cfg, weight_quantizer=wq, input_quantizer=xq,
                                  inputs=cache, eval_inputs=cache, eval_module=_Pair(a, b), eval_kwargs={}, world=world)
    return scale.numpy(), a.weight.data.cpu().numpy()


def _smooth_worker(rank: int, world: int, port: int, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        out.put((rank, [_smooth(world, rank, s) for s in ("Manual", "GridSearch")]))
    finally:
        dist.destroy_process_group()


def test_token_sharded_smoothing_decides_like_the_unsharded_one():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_smooth_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, strategy in enumerate(("Manual", "GridSearch")):
        scale, weight = _smooth(1, 0, strategy)
        assert not np.all(scale == 1.0)
        for r in range(world):  # abs-max spans reduce exactly; the winner of the search is the same
            np.testing.assert_array_equal(results[r][i][0], scale, err_msg=f"{strategy} rank {r}")
            np.testing.assert_array_equal(results[r][i][1], weight, err_msg=f"{strategy} rank {r}")


def _stage(world: int, rank: int, gptq: bool):
    """The weight stage on a toy layer (two linears fed by the same tokens, one by its own), 4 batches of tokens; with
    two ranks each holds two of the batches and half of every linear's output rows."""
    import torch.nn as nn

    from deepcompressor_amd.app import LinearEntry, LlmWeightQuantizerConfig, quantize_layer_weights
    from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                                 SearchBasedCalibStrategy, SkipBasedDynamicRangeCalibConfig)
    from deepcompressor_amd.data.cache import IOTensorsCache, TensorCache, TensorsCache
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer import QuantGptqConfig

    g = torch.Generator().manual_seed(5)
    shapes = [("q_proj", 96, 256, "a"), ("k_proj", 40, 256, "a"), ("down_proj", 64, 384, "b")]
    xs = {}
    for key, c in (("a", 256), ("b", 384)):
        gains = torch.exp(torch.randn(c, generator=g) * 0.7)
        batches = [(torch.randn(1, 160, c, generator=g) * gains).to(torch.bfloat16).cuda() for _ in range(4)]
        mine = batches[rank::world]
        xs[key] = TensorCache(mine, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=len(mine),
                              num_total=len(mine), num_samples=len(mine), orig_device=torch.device("cuda"))
    entries, layer_cache = [], {}
    for name, r, c, key in shapes:
        w = torch.randn(r, c, generator=g) * 0.03
        w = torch.where(torch.rand(r, c, generator=g) < 0.01, w * 5, w).to(torch.bfloat16).cuda()
        module = nn.Linear(c, r, bias=False, device="meta")
        module.weight = nn.Parameter(w, requires_grad=False)
        entries.append(LinearEntry(key=name, name=f"layer.{name}", module=module))
        layer_cache[f"layer.{name}"] = IOTensorsCache(inputs=TensorsCache(xs[key]))
    config = LlmWeightQuantizerConfig(
        dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 128),), scale_dtypes=(torch.float16,),
        calib_range=SkipBasedDynamicRangeCalibConfig(
            degree=2.4, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0, num_grids=20),
        kernel_gptq=QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250) if gptq else None)
    state: dict = {}
    quantize_layer_weights(entries, config, state, layer_cache=layer_cache, streams=2 if gptq else 1, world=world,
                           rank=rank)
    return [e.module.weight.data.float().cpu().numpy() for e in entries]


def _stage_worker(rank: int, world: int, port: int, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        out.put((rank, [_stage(world, rank, gptq) for gptq in (False, True)]))
    finally:
        dist.destroy_process_group()


def test_layer_stage_split_by_tokens_and_rows_equals_the_whole_stage():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stage_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, gptq in enumerate((False, True)):
        whole = _stage(1, 0, gptq)
        for a, b, w in zip(results[0][i], results[1][i], whole):
            np.testing.assert_array_equal(a, b)  # every rank ends with the same full weights
            assert a.shape == w.shape
            if gptq:  # the Hessian's float32 sums run in another order when the tokens are split
                assert (a == w).mean() >= 0.99
            else:  # search + round-to-nearest are independent per row: nothing may differ
                np.testing.assert_array_equal(a, w)
