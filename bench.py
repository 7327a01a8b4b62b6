"""Contract benchmark: quantized params/sec for Llama-3-8B GPTQ W4 g128 + range search (BASELINE.json configs[1]).

One step = the whole weight-quantization hot path for ONE Llama-3-8B decoder layer (7 linears, 218.1 M params)
against 128 x 2048 = 262,144 synthetic calibration tokens already resident in HBM:
  per distinct activation (attention input, o_proj input, MLP input, down_proj input): H = 2/n X^T X   (MFMA)
  per linear: fused 101-candidate range-ratio search -> static ranges -> scales/zeros -> GPTQ sweep -> dequantize.
Layers are independent, so N GPUs each run K layers (weak scaling, no data-path collective).

    python bench.py [--gpus N --steps K --warmup W] [--tokens T] [--shard layers|rows] [--no-cpu-baseline]

``--gpus N`` with N > 1 and no RANK in the environment: this process is only a launcher.  Before anything touches the
GPU it starts N fresh ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one GPU
each, RCCL), relays rank 0's JSON line and exits non-zero if any rank fails.  Under ``python -m torch.distributed.run``
(RANK set) the script is one of the ranks.  ``--shard layers`` (default): every rank quantizes whole layers of its own
(weak scaling, no data-path collective).  ``--shard rows``: ONE layer per step for the whole job -- the calibration
tokens are split over the ranks (partial Hessians all-reduced), every rank searches and sweeps its slab of output rows
of each linear, the slabs are all-gathered (strong scaling).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = None  # imported by main() in the ranks; the launcher never loads it

LLAMA3_8B_LAYER = [  # (name, out_features, in_features, activation id)
    ("q_proj", 4096, 4096, "attn_in"), ("k_proj", 1024, 4096, "attn_in"), ("v_proj", 1024, 4096, "attn_in"),
    ("o_proj", 4096, 4096, "attn_out"), ("gate_proj", 14336, 4096, "mlp_in"), ("up_proj", 14336, 4096, "mlp_in"),
    ("down_proj", 4096, 14336, "mlp_mid"),
]
PARAMS_PER_LAYER = sum(r * c for _, r, c, _ in LLAMA3_8B_LAYER)


def make_layer(device, tokens: int, seed: int, weight_seed: int | None = None):
    """Synthetic bf16 weights and activations of one decoder layer (per-channel log-normal activation gains).
    ``weight_seed``: draw the weights from a generator of their own (ranks that share one layer's weights but hold
    different calibration tokens)."""
    from deepcompressor_amd.data.cache import TensorCache
    from deepcompressor_amd.data.utils.reshape import LinearReshapeFn

    g = torch.Generator(device=device).manual_seed(seed)
    acts, caches = {}, {}
    for _, _, c, key in LLAMA3_8B_LAYER:
        if key in acts:
            continue
        gains = torch.exp(torch.randn(c, generator=g, device=device))
        x = torch.empty((tokens, c), dtype=torch.bfloat16, device=device)
        for t0 in range(0, tokens, 16384):
            n = min(16384, tokens - t0)
            x[t0:t0 + n] = (torch.randn((n, c), generator=g, device=device) * gains).to(torch.bfloat16)
        acts[key] = x
        caches[key] = TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_cached=1, num_total=1,
                                  num_samples=tokens, orig_device=device)
    if weight_seed is not None:
        g = torch.Generator(device=device).manual_seed(weight_seed)
    weights = {name: (torch.randn((r, c), generator=g, device=device) * 0.02).to(torch.bfloat16)
               for name, r, c, _ in LLAMA3_8B_LAYER}
    return weights, caches


def build_objects():
    from deepcompressor_amd.calib import (DynamicRangeCalibConfig, SearchBasedCalibGranularity,
                                          SearchBasedCalibObjective, SearchBasedCalibStrategy)
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer import QuantGptqConfig, Quantizer, QuantizerConfig

    # examples/llm/configs/gptq.yaml of the reference with enable_calib_range = true ("GPTQ-R")
    qcfg = QuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 128),),
                           scale_dtypes=(torch.float16,))
    rcfg = DynamicRangeCalibConfig(degree=2.4, objective=SearchBasedCalibObjective.TensorError,
                                   strategy=SearchBasedCalibStrategy.GridSearch,
                                   granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0,
                                   num_grids=100)
    gcfg = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
    return Quantizer, qcfg, rcfg, gcfg


def stage_config():
    """examples/llm/configs/gptq.yaml of the reference with enable_calib_range = true ("GPTQ-R") as the weight
    stage's config."""
    from deepcompressor_amd.app import LlmWeightQuantizerConfig
    from deepcompressor_amd.calib import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                          SearchBasedCalibStrategy)
    from deepcompressor_amd.calib.config import SkipBasedDynamicRangeCalibConfig
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer import QuantGptqConfig

    return LlmWeightQuantizerConfig(
        dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 128),), scale_dtypes=(torch.float16,),
        calib_range=SkipBasedDynamicRangeCalibConfig(
            degree=2.4, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0, num_grids=100),
        kernel_gptq=QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512))


def layer_entries(weights, caches):
    """The 7 linears of a layer as the weight stage takes them: (entries, layer cache)."""
    import torch.nn as nn

    from deepcompressor_amd.app import LinearEntry
    from deepcompressor_amd.data.cache import IOTensorsCache, TensorsCache

    entries, layer_cache = [], {}
    for name, r, c, key in LLAMA3_8B_LAYER:
        w = weights[name]
        module = nn.Linear(c, r, bias=False, device="meta")
        module.weight = nn.Parameter(w, requires_grad=False)  # the stage replaces .data, `weights` keeps the original
        entries.append(LinearEntry(key=name, name=f"layer.{name}", module=module))
        layer_cache[f"layer.{name}"] = IOTensorsCache(inputs=TensorsCache(caches[key]))
    return entries, layer_cache


def caches_of_another_layer(caches):
    """The same activations as the cached inputs of ANOTHER layer: new tensor objects over the same memory.  The stage
    tells caches apart by the identity of their tensors, as it must between the layers of a model; with these, two
    consecutive steps look to it like two consecutive layers, and nothing computed for one is taken for the other."""
    from deepcompressor_amd.data.cache import TensorCache

    return {key: TensorCache([t.view(t.shape) for t in c.data], channels_dim=c.channels_dim, reshape=c.reshape,
                             num_cached=c.num_cached, num_total=c.num_total, num_samples=c.num_samples,
                             orig_device=c.orig_device) for key, c in caches.items()}


def quantize_layer(weights, caches, config, syrk_events=None, streams: int = 1, world: int = 1, rank: int = 0,
                   next_caches=None):
    """The hot path for one layer through the weight stage of the package (the reference's
    app/llm/quant/weight.py:24-105): range search, then GPTQ fed by the cached inputs, for the 7 linears.  Returns the
    modules holding the dequantized weights (kept so nothing is optimised away).  ``next_caches``: the cached inputs of
    the layer the next call is for; its Hessians are queued beside this layer's pipelines (the stage's ``prefetch``)."""
    from deepcompressor_amd.app import prepare_layer_hessians, quantize_layer_weights

    entries, layer_cache = layer_entries(weights, caches)

    def hessian_hook(channels):
        if syrk_events is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        syrk_events.append((e0, e1, channels))
        return e0, e1  # recorded by the library right around this Hessian's SYRK launches, on their stream

    prefetch = None
    if next_caches is not None:
        next_entries, next_cache = layer_entries(weights, next_caches)
        prefetch = lambda: prepare_layer_hessians(next_entries, next_cache, hessian_hook=hessian_hook)  # noqa: E731
    quantize_layer_weights(entries, config, {}, layer_cache=layer_cache, streams=streams, hessian_hook=hessian_hook,
                           world=world, rank=rank, prefetch=prefetch)
    return [e.module for e in entries]


PMC_FILE = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_stage_pmc.json") for r in (3, 2)) if os.path.exists(p)),
                os.path.join(ROOT, "profiles", "r3_stage_pmc.json"))
SYRK_PMC_FILE = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_syrk_traffic.json") for r in (3, 2))
                      if os.path.exists(p)), os.path.join(ROOT, "profiles", "r3_syrk_traffic.json"))
VALU_ISSUE_PER_S = 256 * 4 * 2.4e9 / 2  # wave-instructions: 4 SIMDs per CU, a wave64 instruction takes 2 cycles


def _pmc(kernel: str) -> dict:
    """Per-launch counter averages of `kernel` from the committed rocprofv3 PMC passes (tools/stage_kernels.py under
    `rocprofv3 --pmc ...`, condensed by tools/kernel_pmc.py); {} when the file is absent."""
    if not os.path.exists(PMC_FILE):
        return {}
    with open(PMC_FILE) as f:
        return json.load(f)["kernels"].get(kernel, {})


def stage_rooflines(device, weights, objs) -> list:
    """Rooflines of the two stage kernels that are not matrix work, measured live outside the timed steps:

    * ``fused_int_kernel``: one fake-quantization pass over a 32768 x 8192 bf16 tensor (1.07 GB, four times the
      Infinity Cache, so every byte comes from and goes to HBM) against its algorithmic 4 B/param (a call that
      returns the fake-quantized tensor only does not write the per-group scales and zeros);
    * ``range_search_kernel``: the 101-candidate search on gate_proj (14336 x 4096): HBM fraction from its 2 B/param +
      the per-group outputs, and the fraction of the chip's vector-instruction issue rate, from the instruction count
      of the committed counter pass (SQ_INSTS_VALU per launch) over the duration measured here.

    Durations are HIP events on the launching stream around back-to-back ``quantize()`` calls (the kernels outlast
    the host's per-call work).  The kernels detect NaN / Inf while they read; in these loops the host's look at their
    flag -- by default a blocking 4-byte read before every return, as the reference asserts on every call -- is made
    once after the timed calls (``engine.FINITE_CHECKS = "deferred"``), so that launches queue back to back and the
    figure is the kernels' rate; ``default_policy_ms`` is the same call with the blocking look per call."""
    from deepcompressor_amd import engine
    from deepcompressor_amd.calib import calibrate_dynamic_range
    from deepcompressor_amd.data.common import TensorType
    import torch.nn as nn

    Quantizer, qcfg, rcfg, _ = objs
    q = Quantizer(qcfg, key="rate", develop_dtype=torch.float32)

    def timed(fn, reps, policy="deferred"):
        before, engine.FINITE_CHECKS = engine.FINITE_CHECKS, policy
        try:
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            engine.raise_if_nonfinite()
        finally:
            engine.FINITE_CHECKS = before
        return e0.elapsed_time(e1) / reps * 1e-3

    out = []
    big = (torch.randn(32768, 8192, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    t_big = timed(lambda: q.quantize(big), 30)
    t_big_default = timed(lambda: q.quantize(big), 10, policy="always")
    big_bytes = big.numel() * 4  # bf16 read + bf16 write (scales / zeros stay on the chip: only the tensor is asked for)
    pmc = _pmc("fused_int_kernel")
    out.append({"kernel": "fused_int_kernel (fake-quant uint4 g128 + zero point, bf16 in / bf16 out)", "bound": "hbm",
                "shape": list(big.shape), "achieved": big_bytes / t_big / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": big_bytes / t_big / 8e12, "frac_of_measured_copy_6290": big_bytes / t_big / 6.29e12,
                "algorithmic_bytes": big_bytes, "traffic": pmc.get("hbm_bytes"), "avg_launch_ms": t_big * 1e3,
                "default_policy_ms": t_big_default * 1e3,
                "valu_insts_per_launch": pmc.get("avg_per_launch", {}).get("SQ_INSTS_VALU")})
    # The other two weight formats of the reference's examples, same tensor: NVFP4 (float4 in groups of 16, float8
    # group scales under one per-tensor scale: a max|x| reduction, then one pass) and QoQ W4 (per-row fp16 scales,
    # then uint4 groups of 128 with integer scales: one pass).  Algorithmic bytes: every read and write the format
    # needs (NVFP4 reads the tensor twice, the per-tensor scale depends on all of it).
    from deepcompressor_amd.data.dtype import QuantDataType
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer import ProgressiveQuantizerConfig, QuantizerConfig

    n_big = big.numel()
    nvfp4 = Quantizer(QuantizerConfig(dtype=QuantDataType.from_str("sfp4_e2m1_all"), zero_point=None,
                                      group_shapes=((-1, -1), (1, 16)),
                                      scale_dtypes=(None, QuantDataType.from_str("sfp8_e4m3_nan"))), key="nvfp4")
    qoq = Quantizer(ProgressiveQuantizerConfig(
        dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale, group_shapes=((1, -1), (1, 128)),
        scale_dtypes=(torch.float16, QuantDataType.from_str("sint8")),
        intermediate_dtypes=(QuantDataType.from_str("sint8"),), intermediate_levels=(0,)), key="qoq")
    for label, quantizer, nbytes, kernels in (
            ("tensor_absmax_kernel + fused_table_kernel (NVFP4 g16: two reads, one write)",
             nvfp4, n_big * 6, ("tensor_absmax_kernel", "fused_table_kernel")),
            ("progressive_int_kernel (QoQ W4 g128: per-row fp16 scale and uint4 groups with integer scales, one pass)",
             qoq, n_big * 4 + big.shape[0] * 4, ("progressive_int_kernel",))):
        t = timed(lambda: quantizer.quantize(big), 30)
        t_default = timed(lambda: quantizer.quantize(big), 10, policy="always")
        counted = [_pmc(k).get("hbm_bytes") for k in kernels]
        out.append({"kernel": label, "bound": "hbm", "shape": list(big.shape), "achieved": nbytes / t / 1e9,
                    "peak": 8000.0, "unit": "GB/s", "frac": nbytes / t / 8e12, "algorithmic_bytes": nbytes,
                    "traffic": sum(counted) if all(c is not None for c in counted) else None,
                    "avg_launch_ms": t * 1e3, "default_policy_ms": t_default * 1e3,
                    "valu_insts_per_launch": sum(_pmc(k).get("avg_per_launch", {}).get("SQ_INSTS_VALU") or 0
                                                 for k in kernels) or None,
                    "note": "duration of the whole quantize() call (all its launches)"})
    del big
    w = weights["gate_proj"]
    t_fq = timed(lambda: q.quantize(w), 50)
    t_fq_default = timed(lambda: q.quantize(w), 20, policy="always")
    n = w.numel()
    fq_bytes = n * 4
    out.append({"kernel": "fused_int_kernel on gate_proj (235 MB working set: Infinity-Cache resident when repeated)",
                "bound": "hbm", "shape": list(w.shape), "achieved": fq_bytes / t_fq / 1e9, "peak": 8000.0,
                "unit": "GB/s", "frac": fq_bytes / t_fq / 8e12, "algorithmic_bytes": fq_bytes, "traffic": None,
                "avg_launch_ms": t_fq * 1e3, "default_policy_ms": t_fq_default * 1e3})
    t_rs = timed(lambda: calibrate_dynamic_range(TensorType.Weights, rcfg, True, q, modules=[None], activations=None,
                                                 weights=[nn.Parameter(w, requires_grad=False)]), 3)
    rs_bytes = n * 2 + (n // 128) * 16  # bf16 read once; max, min, ratio, error per group
    pmc = _pmc("range_search_kernel")
    counters = pmc.get("avg_per_launch", {})
    insts = counters.get("SQ_INSTS_VALU")
    # share of the cycles in which a SIMD's vector pipe is executing, from the counter pass itself:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
    busy = None
    if counters.get("SQ_ACTIVE_INST_VALU") and counters.get("GRBM_GUI_ACTIVE"):
        busy = counters["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * counters["GRBM_GUI_ACTIVE"] / 8)
    out.append({"kernel": "range_search_kernel (101 candidates per group, one pass over the weight)", "bound": "valu",
                "shape": list(w.shape), "candidates": 101, "avg_launch_ms": t_rs * 1e3,
                "Gparam_per_s": n / t_rs / 1e9, "Gcandidate_evals_per_s": n * 101 / t_rs / 1e9,
                "hbm": {"achieved": rs_bytes / t_rs / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": rs_bytes / t_rs / 8e12, "algorithmic_bytes": rs_bytes, "traffic": pmc.get("hbm_bytes")},
                "valu": {"insts_per_launch": insts, "peak_wave_insts_per_s": VALU_ISSUE_PER_S,
                         "achieved_wave_insts_per_s": insts / t_rs if insts else None,
                         "frac": busy, "issue_frac_at_2_cycles_per_inst": insts / t_rs / VALU_ISSUE_PER_S if insts else None,
                         "note": "frac = vector-pipe busy share of the kernel's cycles in the committed counter pass "
                                 "(packed two-value float32 instructions occupy the pipe for 4 cycles, twice the "
                                 "2-cycle issue slot the second figure assumes)",
                         "insts_per_candidate_eval": insts * 64 / (n * 101) if insts else None}})
    return out


def other_configs(device) -> list:
    """Wall time of one unit of the other workloads the baseline file lists (they are parity-test cases; the headline
    value above is the GPTQ one).  Synthetic inputs resident in HBM, one warm-up call, then the mean of a few calls
    bracketed by device synchronisation."""
    import torch.nn as nn

    from deepcompressor_amd.calib import calibrate_dynamic_range
    from deepcompressor_amd.calib.config import (DynamicRangeCalibConfig, SearchBasedCalibGranularity,
                                                 SearchBasedCalibObjective, SearchBasedCalibStrategy, SmoothCalibConfig,
                                                 SmoothSpanMode)
    from deepcompressor_amd.calib.smooth import SmoothLinearCalibrator
    from deepcompressor_amd.data.cache import TensorCache, TensorsCache
    from deepcompressor_amd.data.common import TensorType
    from deepcompressor_amd.data.dtype import QDType, QuantDataType
    from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer import ProgressiveQuantizerConfig, Quantizer, QuantizerConfig
    from deepcompressor_amd.quantizer.config.lowrank import QuantLowRankConfig

    def wall(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    g = torch.Generator(device=device).manual_seed(99)
    rand = lambda *shape: torch.randn(*shape, generator=g, device=device)  # noqa: E731
    out = []
    # configs[0]: sint4 g128 fake-quant + range-ratio search on a 4096 x 4096 bf16 linear
    w = (rand(4096, 4096) * 0.02).to(torch.bfloat16)
    q0 = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 128),),
                                   scale_dtypes=(torch.float16,)), key="c0")
    rcfg = DynamicRangeCalibConfig(objective=SearchBasedCalibObjective.TensorError,
                                   strategy=SearchBasedCalibStrategy.GridSearch,
                                   granularity=SearchBasedCalibGranularity.Group, degree=2, ratio=1.0, max_shrink=0.2,
                                   max_expand=1.0, num_grids=80)

    def c0():
        dyn = calibrate_dynamic_range(TensorType.Weights, rcfg, True, q0, modules=[None], activations=None,
                                      weights=[nn.Parameter(w, requires_grad=False)])
        return q0.quantize(w, dynamic_range=dyn)

    t = wall(c0, 5)
    out.append({"config": "configs[0]: sint4 g128 range-ratio search (81 candidates) + fake-quant, 4096 x 4096 bf16",
                "ms": t * 1e3, "Gparam_per_s": w.numel() / t / 1e9})
    # configs[2]: QoQ W4 (per-channel sint8 under fp16 scales, then uint4 g128 with int8 scales) on the 7 linears of a
    # Llama-3-8B layer
    qoq = Quantizer(ProgressiveQuantizerConfig(
        dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale, group_shapes=((1, -1), (1, 128)),
        scale_dtypes=(torch.float16, QuantDataType.from_str("sint8")),
        intermediate_dtypes=(QuantDataType.from_str("sint8"),), intermediate_levels=(0,)), key="c2")
    layer = [(rand(r, c) * 0.02).to(torch.float16) for _, r, c, _ in LLAMA3_8B_LAYER]
    from deepcompressor_amd import engine

    t = wall(lambda: [qoq.quantize(x) for x in layer], 5)
    engine.FINITE_CHECKS = "deferred"  # the NaN / Inf flag looked at once per layer instead of once per linear
    try:
        t_deferred = wall(lambda: ([qoq.quantize(x) for x in layer], engine.raise_if_nonfinite()), 5)
    finally:
        engine.FINITE_CHECKS = "always"
    out.append({"config": "configs[2]: QoQ W4 progressive fake-quant of the 7 linears of one Llama-3-8B layer, fp16",
                "ms": t * 1e3, "Gparam_per_s": PARAMS_PER_LAYER / t / 1e9,
                "ms_with_one_nonfinite_check_per_layer": t_deferred * 1e3,
                "Gparam_per_s_with_one_check_per_layer": PARAMS_PER_LAYER / t_deferred / 1e9})
    del layer
    # configs[3]: SVDQuant W4 (sint4 g64) with a rank-32 low-rank branch on a FLUX.1-dev 3072 x 3072 linear
    svd = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 64),),
                                    scale_dtypes=(torch.float16,)), key="c3", low_rank=QuantLowRankConfig(rank=32))
    wf = (rand(3072, 3072) * 0.02).to(torch.bfloat16)
    t = wall(lambda: svd.quantize_with_low_rank([wf]), 3)
    out.append({"config": "configs[3]: SVDQuant sint4 g64 + rank-32 branch (SVD, residual fake-quant), 3072 x 3072 bf16",
                "ms": t * 1e3, "Gparam_per_s": wf.numel() / t / 1e9})
    del wf
    # configs[4]: SAM2 Hiera-L fc1 (1152 -> 4608) W8A8 per channel / per token, per-channel smoothing search with
    # alpha = 0.5, beta = -2, 20 grids (39 candidates) scored on the layer's outputs, 4096 tokens
    fc1 = nn.Linear(1152, 4608, bias=True, device=device, dtype=torch.float32)
    fc1.weight.data = rand(4608, 1152) * 0.03
    gains = torch.exp(rand(1152) * 1.2)
    xs = [rand(1, 1024, 1152) * gains for _ in range(4)]
    cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=4, num_total=4,
                                     num_samples=4, orig_device=device))
    w8 = QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(None,))
    wq, xq = Quantizer(w8, key="w"), Quantizer(w8, channels_dim=-1, key="x")
    scfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=20,
                             strategy=SearchBasedCalibStrategy.GridSearch,
                             objective=SearchBasedCalibObjective.OutputsError,
                             granularity=SearchBasedCalibGranularity.Layer)

    def c4():
        return SmoothLinearCalibrator(scfg, wq, xq).calibrate(x_wgts=[fc1.weight], x_acts=cache, x_mods=[fc1],
                                                              eval_inputs=cache, eval_module=fc1, eval_kwargs={})

    t = wall(c4, 3)
    out.append({"config": "configs[4]: Hiera-L fc1 1152 -> 4608 W8A8 smoothing search, 39 candidates, 4096 tokens, f32",
                "ms": t * 1e3, "candidates_per_s": 39 / t})
    return out


def cpu_baseline(w_gpu, x_gpu, tokens: int):
    """The CPU oracle (numpy restatement of the reference) on a bounded sample of the headline workload: ONE of the
    step's seven linears, k_proj (1024 x 4096), with the step's own weights and ALL of its calibration tokens -- the
    Hessian of 262,144 x 4096 activations in the reference's 512-row blocks (float32 BLAS on every host thread), the
    101-ratio search with the weight's rows dealt over a thread pool, the inverse factor (LAPACK) and the column
    sweep.  Returns the cpu_baseline object of the bench contract."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    import gptq_oracle as go
    import quant_oracle as qo

    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    w = w_gpu.float().cpu().numpy()
    x = x_gpu.float().cpu().numpy()  # (tokens, 4096) float32 holding bf16 values
    R, C = w.shape
    spec = qo.spec_from_case(dict(dtype="uint4", zero="PostScale", groups=((1, 128),), sdt=("f16",)))
    ratios = qo.range_ratios(100, 0.2, 1.0)
    t0 = time.perf_counter()
    # per-group search: rows are independent, one slab per thread (numpy releases the interpreter lock in its loops)
    slabs = np.array_split(np.arange(R), min(threads, R // 16))
    with ThreadPoolExecutor(max_workers=len(slabs)) as pool:
        parts = list(pool.map(lambda rows: qo.range_search_group(w[rows], "bf16", spec, ratios, 2.4), slabs))
    best = {k: np.concatenate([p[k] for p in parts], axis=0) for k in ("max", "min")}
    t_search = time.perf_counter() - t0
    base = qo.fake_quantize(w, "bf16", spec, dynamic_range=[{"max": best["max"], "min": best["min"]}])
    t1 = time.perf_counter()
    H = go.hessian(x, tokens, 512)
    t_hessian = time.perf_counter() - t1
    t2 = time.perf_counter()
    go.gptq_codes(w, None, tokens, spec.steps[0], base["scale"], base["zero"], base["view_shape"], H=H)
    t_sweep = time.perf_counter() - t2
    total = time.perf_counter() - t0
    return {
        "value": R * C / total, "unit": "params/s", "cores": threads, "kind": "port",
        "sample": (f"k_proj {R}x{C} bf16 of the timed step with all {tokens} calibration tokens: 101-ratio range search "
                   f"({t_search:.1f} s, rows over {len(slabs)} threads), Hessian in 512-row blocks ({t_hessian:.1f} s, "
                   f"{2.0 * tokens * C * C / t_hessian / 1e9:.0f} GFLOP/s float32 BLAS), inverse factor + column sweep "
                   f"({t_sweep:.1f} s); numpy oracle on {threads} threads.  One of the step's seven linears (1.9 % of "
                   "its parameters; q/k/v share this Hessian, the reference recomputes it per linear).  The "
                   "reference's own torch-CPU path, timed in the build container on 8 threads: range search 2.10 "
                   "Mparam/s, GPTQ 4096x4096 with 8192 tokens 4.68 Mparam/s, one fake-quant pass 95.6 Mparam/s -- a "
                   "reported baseline only, not a basis for speed-up claims"),
        "seconds": {"search": t_search, "hessian": t_hessian, "inverse_factor_and_sweep": t_sweep, "total": total},
        "reference_torch_cpu_8_threads": {"range_search_Mparam_per_s": 2.10, "gptq_8192_tokens_Mparam_per_s": 4.68,
                                          "fake_quant_Mparam_per_s": 95.6, "where": "build container, 8 vCPU"},
    }


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch(n: int, argv: list, dry: bool) -> int:
    """Start ``n`` ranks of this script, one per GPU, and relay rank 0's output.  Runs in a process that has not
    loaded torch or touched the GPU; the ranks are fresh child processes (never an exec of an initialised one).
    Returns the exit code: 0 only when every rank exited with 0."""
    import subprocess

    port = _free_port()
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), *argv], env=env,
                                         stdout=subprocess.PIPE if (r == 0 or dry) else sys.stderr, text=True))
    failed = 0
    if dry:
        seen = []
        for child in children:
            text, _ = child.communicate()
            failed = failed or child.returncode
            seen.extend(json.loads(line) for line in text.splitlines() if line.startswith("{"))
        print(json.dumps({"dry_launch": seen}))
        return failed
    import threading

    def relay() -> None:
        for line in children[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    pending = set(range(n))
    while pending and not failed:
        for r in sorted(pending):
            code = children[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    failed = code
                    print(f"bench.py: rank {r} exited with {code}", file=sys.stderr)
        time.sleep(0.2)
    if failed:  # a rank died: the others would wait in a collective for ever
        for r in pending:
            children[r].terminate()
        for r in pending:
            try:
                children[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                children[r].kill()
    pump.join(timeout=10)
    return failed


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tokens", type=int, default=128 * 2048, help="calibration tokens (contract value: 262144)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage rooflines and the other configs")
    ap.add_argument("--streams", type=int, default=4,
                    help="run the independent groups of linears of a layer concurrently on this many HIP streams")
    ap.add_argument("--pipeline", action="store_true",
                    help="queue the next step's Hessians beside the current step's factorisations and sweeps (the "
                         "stage's `prefetch`; measured: no gain, the step is bound by the kernels that fill the chip)")
    ap.add_argument("--shard", choices=("layers", "rows"), default="layers",
                    help="layers: every rank quantizes whole layers of its own (weak scaling); rows: one layer per "
                         "step for the whole job, tokens and output rows split over the ranks (strong scaling)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the ranks (nccl = RCCL); gloo is "
                                                      "for rehearsals with several ranks on one GPU")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, have each report its environment and exit (no GPU work)")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.dry_launch):
        sys.exit(launch(args.gpus, sys.argv[1:], args.dry_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_launch:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world_size": world, "pid": os.getpid(),
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}))
        return
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size counts", file=sys.stderr)
    global torch
    import torch

    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        index = local_rank % max(1, torch.cuda.device_count()) if args.backend != "nccl" else local_rank
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", index))
        else:
            dist.init_process_group(backend=args.backend)
        world = dist.get_world_size()  # what the communicator saw, reported as n_gpus
    else:
        index = local_rank
    device = torch.device("cuda", index)
    torch.cuda.set_device(device)

    from deepcompressor_amd.parallel import barrier_and_max

    rows_mode = args.shard == "rows" and world > 1
    objs = build_objects()
    config = stage_config()
    if rows_mode:
        # one layer for the whole job: the same weights on every rank, an equal share of the tokens each
        assert args.tokens % world == 0, "--shard rows: the token count must divide by the number of ranks"
        my_tokens, split = args.tokens // world, dict(world=world, rank=rank)
        weights, caches = make_layer(device, my_tokens, seed=1234 + rank, weight_seed=4321)
    else:
        my_tokens, split = args.tokens, {}
        weights, caches = make_layer(device, args.tokens, seed=1234 + rank)
    # --pipeline: steps are consecutive layers, and while a layer's factorisations and sweeps run, the next layer's
    # Hessians are queued behind its own (the stage's `prefetch`).  Every step still computes its four Hessians itself,
    # inside the loop it is counted in: the last step of either loop prepares nothing for a step after it, and the two
    # sets of cache objects (the same activations) keep the stage from taking one step's Hessians for another's.
    pipelined = args.pipeline and not rows_mode and args.streams > 1
    cache_sets = [caches, caches_of_another_layer(caches)] if pipelined else [caches, caches]

    def run_steps(n, events=None):
        out = None
        for i in range(n):
            ahead = cache_sets[(i + 1) % 2] if pipelined and i + 1 < n else None
            out = quantize_layer(weights, cache_sets[i % 2], config, events, streams=args.streams, next_caches=ahead,
                                 **split)
        return out

    run_steps(args.warmup)
    torch.cuda.synchronize()
    barrier_and_max(0.0, world, device)
    syrk_events: list = []
    t0 = time.perf_counter()
    outputs = run_steps(args.steps, syrk_events)
    torch.cuda.synchronize()
    elapsed = barrier_and_max(time.perf_counter() - t0, world, device)
    del outputs
    # the same four Hessians once more with nothing else on the chip (outside the timed region): in the timed steps
    # the three smaller launches overlap the kernels of the largest group's pipeline
    alone_events: list = []
    if rank == 0:
        from deepcompressor_amd import _lib
        from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod

        seen = set()
        for _, _, c, key in sorted(LLAMA3_8B_LAYER, key=lambda row: -row[2]):  # largest first, as in the step
            if key in seen:
                continue
            seen.add(key)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            alone_events.append((e0, e1, c))
            gptq_mod.prepare_hessians([caches[key]], device, profile_events=(e0, e1))
        torch.cuda.synchronize()
        gptq_mod.clear_hessian_cache()

    if rank == 0:
        total_params = PARAMS_PER_LAYER * args.steps * (1 if rows_mode else world)
        # dominant kernel: the MFMA SYRK.  Algorithmic work of one launch = 2 * T * C^2 flop (X^T X as the reference
        # computes it); the kernel executes the upper-triangular tiles only.
        syrk_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in syrk_events)
        syrk_flop = sum(2.0 * my_tokens * c * c for _, _, c in syrk_events)
        tiles = lambda c: (-(-c // 256)) * (-(-c // 256) + 1) // 2  # noqa: E731
        exec_flop = sum(2.0 * my_tokens * tiles(c) * 256 * 256 for _, _, c in syrk_events)
        algorithmic = syrk_flop / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        executed = exec_flop / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        # HBM-side bytes per launch of that kernel, from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, see the
        # file for the method); only valid for the contract workload it was measured on
        traffic = None
        tpath = SYRK_PMC_FILE if os.path.exists(SYRK_PMC_FILE) else os.path.join(ROOT, "profiles", "r1_syrk_traffic.json")
        if my_tokens == 128 * 2048 and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f)["avg_hbm_bytes_per_launch"]
        alone_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in alone_events)
        alone_exec = sum(2.0 * my_tokens * tiles(c) * 256 * 256 for _, _, c in alone_events)
        alone = alone_exec / (alone_ms * 1e-3) / 1e12 if alone_ms > 0 else 0.0
        n_launch = max(1, len(syrk_events))
        # algorithmic bytes of a launch: X read once (2 T C) and H written once (4 C^2)
        syrk_bytes = sum(2.0 * my_tokens * c + 4.0 * c * c for _, _, c in syrk_events) / n_launch
        line = {
            "metric": "quantized params/sec (whole node), Llama-3-8B GPTQ W4 g128 + range search",
            "value": total_params / elapsed, "unit": "params/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if rows_mode else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "GPTQ W4A16 g128 (uint4 PostScale, fp16 scales) + 101-ratio range search on the 7 "
                                   "linears of one Llama-3-8B decoder layer per step, bf16 weights/activations, "
                                   f"{args.tokens} calibration tokens resident in HBM",
                       "params_per_step": PARAMS_PER_LAYER, "tokens": args.tokens,
                       "parallelism": (f"one layer over {world} ranks: tokens split (Hessians all-reduced), output "
                                       f"rows split (slabs all-gathered)" if rows_mode else
                                       f"independent layers x{world}"),
                       "backend": args.backend if world > 1 else None, "streams_per_gpu": args.streams,
                       "next_layer_hessians_prefetched": pipelined},
            # frac = matrix-core utilisation: the flops the kernel EXECUTES (upper-triangular 256-tiles only, about
            # half of 2 T C^2) over the dense bf16 peak; the algorithmic figure (2 T C^2 credited per launch) beside it
            "roofline": {"bound": "mfma", "achieved": executed, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": executed / 2500.0, "algorithmic_achieved": algorithmic,
                         "algorithmic_frac": algorithmic / 2500.0, "traffic": traffic,
                         "algorithmic_bytes": syrk_bytes,
                         "kernel": "syrk_kernel (H += 2/n X^T X, bf16 MFMA)",
                         "launches": len(syrk_events), "avg_launch_ms": syrk_ms / n_launch,
                         "alone": {"achieved": alone, "frac": alone / 2500.0, "launches": len(alone_events),
                                   "avg_launch_ms": alone_ms / max(1, len(alone_events)),
                                   "note": "the same launches with nothing else on the chip, outside the timed "
                                           "region; inside it the smaller Hessians overlap the largest group's "
                                           "factorisation and sweeps"},
                         "share_of_step": syrk_ms * 1e-3 / elapsed},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights["k_proj"], caches["attn_in"].data[0], my_tokens)
        if world == 1 and not args.no_extras:
            line["rooflines"] = stage_rooflines(device, weights, objs)
            del weights, caches
            line["other_configs"] = other_configs(device)
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
