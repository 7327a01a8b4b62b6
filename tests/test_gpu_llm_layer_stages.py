"""Rotation and smoothing stages on the GPU (``rotate_llm``, ``smooth_llm_layer``) against the reference's own stage
drivers run on a two-layer random Llama.

Bars: rotation -- every parameter within 1 bfloat16 ulp of the reference's, >= 99.9 % identical (float64 products
summed in another order); smoothing scales <= 4 ulp (quotients of float32 powers); parameters after manual smoothing
identical wherever the scale is; the grid search picks the reference's candidates (scales <= 4 ulp)."""

import numpy as np
import pytest
import torch
from _tiny_llama import bf16_from_bits, bf16_values, build, state_of
from test_oracle_llm_layer_stages import LS, ulp_bf16, values

from deepcompressor_amd.app import (LlmActivationQuantizerConfig, LlmModelStruct, LlmQuantConfig,
                                    LlmWeightQuantizerConfig, rotate_llm, smooth_llm_layer)
from deepcompressor_amd.calib.config import (QuantRotationConfig, SearchBasedCalibStrategy, SkipBasedSmoothCalibConfig,
                                             SmoothAttentionCalibConfig, SmoothSpanMode, SmoothTransfomerConfig)
from deepcompressor_amd.data.cache import IOTensorsCache, TensorCache
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.data.zero import ZeroPointDomain

pytestmark = pytest.mark.gpu
pytest.importorskip("transformers")


def host_state(model):
    return {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items() if "rotary" not in k}


def test_rotation_stage_matches_reference():
    model = build(LS, "init", "cuda")
    tokens = torch.from_numpy(LS["tokens"][0]).cuda()
    rotation = rotate_llm(model, QuantRotationConfig(random=False, transforms=["out_proj"]))
    np.testing.assert_array_equal(rotation.cpu().numpy(), LS["rotation"])
    got, want = host_state(model), values("rotated")
    assert sorted(got) == sorted(want)
    for k in want:
        assert model.state_dict()[k].dtype == torch.bfloat16, k
        u = ulp_bf16(got[k], want[k])
        assert u.max() <= 1 and (u == 0).mean() >= 0.999, (k, u.max(), (u == 0).mean())
    with torch.no_grad():
        logits = model(tokens).logits.float().cpu().numpy()
    assert np.abs(logits - LS["logits|init"]).max() <= 0.03 * np.abs(LS["logits|init"]).max()  # the function is kept


def smoothing_inputs(model):
    struct = LlmModelStruct.construct(model)
    layer = struct.backbone_struct.layer_structs[0]

    def cache(name, side):
        ts = [bf16_from_bits(a).cuda() for a in LS[f"acts|{name}|{side}"]]
        return TensorCache(ts, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=len(ts), num_total=len(ts),
                           num_samples=len(ts) * ts[0].shape[0], orig_device=torch.device("cuda"))

    layer_cache = {name: IOTensorsCache(inputs=cache(name, "inputs"), outputs=cache(name, "outputs"))
                   for name in LS.meta["watched"]}
    bf = lambda key: torch.from_numpy(LS[key]).to(torch.bfloat16).cuda()  # noqa: E731
    kwargs = dict(attention_mask=bf("kwargs|attention_mask"), position_embeddings=(bf("kwargs|cos"), bf("kwargs|sin")))
    return layer, layer_cache, kwargs


def quant_config(proj, attn):
    f16 = (torch.float16,)
    return LlmQuantConfig(
        wgts=LlmWeightQuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 16),),
                                      scale_dtypes=f16),
        ipts=LlmActivationQuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=f16),
        opts=LlmActivationQuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale,
                                          group_shapes=((1, 16),), scale_dtypes=f16, skips=["attn_q"]),
        smooth=SmoothTransfomerConfig(proj=proj, attn=attn))


SPANS = [(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)]


def check_scales(tag, cache):
    assert sorted(cache) == LS.meta[f"smooth|{tag}"]["scales"]
    for k, v in cache.items():
        want = LS[f"smooth|{tag}|scale|{k}"]
        assert v.device.type == "cpu" and v.dtype == torch.float32
        assert (np.abs(v.numpy().astype(np.float64) - want) <= 4 * np.spacing(want)).all(), k


def test_manual_smoothing_stage_matches_reference():
    model = build(LS, "rotated", "cuda")
    layer, layer_cache, kwargs = smoothing_inputs(model)
    config = quant_config(
        SkipBasedSmoothCalibConfig(strategy=SearchBasedCalibStrategy.Manual, spans=SPANS, alpha=0.3, beta=0.7, skips=[]),
        SmoothAttentionCalibConfig(strategy=SearchBasedCalibStrategy.Manual, spans=SPANS, alpha=0.5, beta=0))
    cache: dict[str, torch.Tensor] = {}
    smooth_llm_layer(layer, config, cache, layer_cache=layer_cache, layer_kwargs=kwargs)
    check_scales("manual", cache)
    got, want = host_state(model), values("smooth|manual|state")
    for k in want:
        u = ulp_bf16(got[k], want[k])  # a scale 1 ulp off moves a scaled weight onto the next bfloat16 value at most
        assert u.max() <= 1 and (u == 0).mean() >= 0.97, (k, u.max(), (u == 0).mean())
    # applying the stored scales to a fresh copy needs no activations and gives the same parameters
    again = build(LS, "rotated", "cuda")
    smooth_llm_layer(LlmModelStruct.construct(again).backbone_struct.layer_structs[0], config, dict(cache))
    for k, v in host_state(again).items():
        np.testing.assert_array_equal(v, got[k], err_msg=k)
    tokens = torch.from_numpy(LS["tokens"][0]).cuda()
    with torch.no_grad():
        logits = model(tokens).logits.float().cpu().numpy()
    assert np.abs(logits - LS["logits|rotated"]).max() <= 0.03 * np.abs(LS["logits|rotated"]).max()


def test_grid_smoothing_stage_picks_the_reference_candidates():
    model = build(LS, "rotated", "cuda")
    layer, layer_cache, kwargs = smoothing_inputs(model)
    config = quant_config(
        SkipBasedSmoothCalibConfig(strategy=SearchBasedCalibStrategy.GridSearch, spans=SPANS, alpha=0.5, beta=-2,
                                   num_grids=4, skips=["qkv_proj", "up_proj"]), None)
    cache: dict[str, torch.Tensor] = {}
    smooth_llm_layer(layer, config, cache, layer_cache=layer_cache, layer_kwargs=kwargs)
    check_scales("grid", cache)
    got, want = host_state(model), values("smooth|grid|state")
    for k in want:
        u = ulp_bf16(got[k], want[k])
        assert u.max() <= 1 and (u == 0).mean() >= 0.97, (k, u.max(), (u == 0).mean())


def test_manual_reorder_stage_matches_reference():
    from deepcompressor_amd.app import reorder_llm_layer
    from deepcompressor_amd.calib.config import ChannelOrderCalibConfig, SkipBasedChannelOrderConfig

    model = build(LS, "rotated", "cuda")
    layer, layer_cache, kwargs = smoothing_inputs(model)
    config = quant_config(None, None)
    config.reorder = SkipBasedChannelOrderConfig(
        skips=["residual"], strategy=SearchBasedCalibStrategy.Manual,
        channel_metric=ChannelOrderCalibConfig.ChannelMetric.InputsAbsMax,
        channel_index=ChannelOrderCalibConfig.ChannelIndex.Sequential)
    cache: dict[str, torch.Tensor] = {}
    leftover = reorder_llm_layer(layer, config, cache, layer_cache=layer_cache, layer_kwargs=kwargs)
    assert leftover is None and sorted(cache) == LS.meta["reorder|manual"]["indexes"]
    import reorder_oracle as rr
    from test_oracle_llm_layer_stages import assert_same_sort, v_proj_order

    rotated, got = values("rotated"), host_state(model)
    p = "model.layers.0"
    for k, v in cache.items():
        assert v.device.type == "cpu" and v.dtype == torch.int64
        # tied abs-max values (bfloat16 activations) may come out in either order: the reference's argsort is not stable
        xs = list(bf16_values(LS[f"acts|{k}|inputs"]))
        metric = rr.channel_metric(xs, [rotated[f"{k}.weight"]], "InputsAbsMax", v.numel())
        assert_same_sort(metric, v.numpy(), LS[f"reorder|manual|index|{k}"])
    o_index, d_index = cache[f"{p}.self_attn.o_proj"].numpy(), cache[f"{p}.mlp.down_proj"].numpy()
    expect = dict(rotated)
    expect[f"{p}.self_attn.o_proj.weight"] = rotated[f"{p}.self_attn.o_proj.weight"][:, o_index]
    expect[f"{p}.self_attn.v_proj.weight"] = rotated[f"{p}.self_attn.v_proj.weight"][v_proj_order(o_index)]
    expect[f"{p}.mlp.down_proj.weight"] = rotated[f"{p}.mlp.down_proj.weight"][:, d_index]
    for n in ("up_proj", "gate_proj"):
        expect[f"{p}.mlp.{n}.weight"] = rotated[f"{p}.mlp.{n}.weight"][d_index]
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=k)
    # with the reference's orders handed in, the parameters are the reference's
    ref_run = build(LS, "rotated", "cuda")
    ref_cache = {k: torch.from_numpy(LS[f"reorder|manual|index|{k}"]) for k in cache}
    reorder_llm_layer(LlmModelStruct.construct(ref_run).backbone_struct.layer_structs[0], config, ref_cache)
    want = values("reorder|manual|state")
    for k, v in host_state(ref_run).items():
        np.testing.assert_array_equal(v, want[k], err_msg=k)
    # a permutation of hidden channels between producer and consumer keeps the function
    tokens = torch.from_numpy(LS["tokens"][0]).cuda()
    with torch.no_grad():
        logits = model(tokens).logits.float().cpu().numpy()
    assert np.abs(logits - LS["logits|rotated"]).max() <= 0.03 * np.abs(LS["logits|rotated"]).max()
    # stored orders are applied as they are
    again = build(LS, "rotated", "cuda")
    reorder_llm_layer(LlmModelStruct.construct(again).backbone_struct.layer_structs[0], config, dict(cache))
    for k, v in host_state(again).items():
        np.testing.assert_array_equal(v, got[k], err_msg=k)


def test_weight_and_activation_stages_after_smoothing_match_reference():
    """The QoQ-style chain's last two stages over the layer struct: progressive weights (sint8 per channel, then
    uint4 groups of 16 with integer scales and a range search per step) -- every scale.pt entry and every dequantized
    weight bit-exact -- then dynamic sint8 inputs / uint4 keys and values hooked onto the layer: the logits with every
    quantizer in place agree with the reference's to bf16 noise."""
    from deepcompressor_amd.app import quantize_llm_layer_activations, quantize_llm_layer_weights
    from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                                 SkipBasedDynamicRangeCalibConfig)

    model = build(LS, "smooth|manual|state", "cuda")
    layer, layer_cache, kwargs = smoothing_inputs(model)
    config = quant_config(None, None)
    config.wgts = LlmWeightQuantizerConfig(
        dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, -1), (1, 16)),
        scale_dtypes=(torch.float16, QDType.sint8), intermediate_dtypes=(QDType.sint8,), intermediate_levels=(0,),
        calib_range=SkipBasedDynamicRangeCalibConfig(
            degree=2, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Group, max_shrink=0.4, max_expand=1.0, num_grids=8, skips=[]))
    weight_ranges: dict = {}
    scale_sd = quantize_llm_layer_weights(layer, config, weight_ranges, layer_cache=layer_cache, layer_kwargs=kwargs,
                                          return_with_scale_state_dict=True)
    assert list(weight_ranges) == LS.meta["chain"]["weight_ranges_in_order"]
    assert sorted(scale_sd) == LS.meta["chain"]["scales"]
    for k, v in scale_sd.items():
        assert v.device.type == "cpu"
        np.testing.assert_array_equal(v.float().numpy().reshape(-1), LS[f"chain|scale|{k}"].reshape(-1), err_msg=k)
    got, want = host_state(model), values("chain|state")
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    act_ranges: dict = {}
    hooked = quantize_llm_layer_activations(layer, config, act_ranges, layer_cache=layer_cache, layer_kwargs=kwargs)
    assert sorted(act_ranges) == LS.meta["chain"]["act_ranges"] == sorted(hooked)
    tokens = torch.from_numpy(LS["tokens"][0]).cuda()
    with torch.no_grad():
        logits = model(tokens).logits.float().cpu().numpy()
    ref = LS["logits|chain"]
    assert np.abs(logits - ref).max() <= 0.03 * np.abs(ref).max()
    # and the quantizers do act: without them the logits sit further from the reference's than with them
    for _, hooks in hooked.values():
        for h in hooks:
            h.remove()
    with torch.no_grad():
        plain = model(tokens).logits.float().cpu().numpy()
    assert np.abs(plain - ref).mean() > np.abs(logits - ref).mean()


def test_gptq_weight_stage_on_the_layer_matches_reference():
    """The weight stage with the GPTQ kernel fed by the layer's cached inputs (64 tokens: the Hessians of the 128-wide
    down projection are rank deficient, only the damping makes them invertible).  The searched scales / zeros do not
    depend on GPTQ and are bit-exact; the GPTQ-updated weights are held to the stated bar: identical codes for most
    weights and layer outputs within 1e-3 of the output's size of the reference's."""
    from deepcompressor_amd.app import quantize_llm_layer_weights
    from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                                 SkipBasedDynamicRangeCalibConfig)
    from deepcompressor_amd.quantizer import QuantGptqConfig

    model = build(LS, "rotated", "cuda")
    layer, layer_cache, kwargs = smoothing_inputs(model)
    config = quant_config(None, None)
    config.wgts = LlmWeightQuantizerConfig(
        dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 16),), scale_dtypes=(torch.float16,),
        calib_range=SkipBasedDynamicRangeCalibConfig(
            degree=2, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Group, max_shrink=0.4, max_expand=1.0, num_grids=8, skips=[]),
        kernel_gptq=QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512))
    scale_sd = quantize_llm_layer_weights(layer, config, {}, layer_cache=layer_cache, layer_kwargs=kwargs,
                                          return_with_scale_state_dict=True)
    assert sorted(scale_sd) == LS.meta["gptq"]["scales"]
    for k, v in scale_sd.items():
        np.testing.assert_array_equal(v.float().numpy().reshape(-1), LS[f"gptq|scale|{k}"].reshape(-1), err_msg=k)
    got, want = host_state(model), values("gptq|state")
    report = {}
    for _, name, module, _, _ in layer.named_key_modules():
        k = f"{name}.weight"
        scale = LS[f"gptq|scale|{k}.scale.0"].reshape(got[k].shape[0], -1, 1)
        steps = np.abs(got[k] - want[k]).reshape(got[k].shape[0], -1, 16) / scale  # differences in quantization steps
        same = float((steps < 0.5).mean())
        xs = [bf16_values(a).reshape(-1, got[k].shape[1]) for a in LS[f"acts|{name}|inputs"]]
        x = np.concatenate(xs).astype(np.float64)
        y_got, y_want = x @ got[k].astype(np.float64).T, x @ want[k].astype(np.float64).T
        out_err = float(np.abs(y_got - y_want).max() / np.abs(y_want).max())
        report[name] = (round(same, 4), out_err)
    print(report)
    for name, (same, out_err) in report.items():
        assert same >= 0.99 and out_err <= 1e-3, report


def test_static_activation_ranges_on_the_layer_match_reference():
    """Static per-tensor sint8 inputs with a range search scored on module outputs (attention block with its keyword
    arguments, the linears themselves, the mlp): the stored ranges are the reference's.  The candidates' errors are
    sums over bf16 module outputs, so a search between near-equal candidates could land on the neighbour; the fixture's
    four searches are not that close."""
    from deepcompressor_amd.app import quantize_llm_layer_activations
    from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                                 SkipBasedDynamicRangeCalibConfig)

    model = build(LS, "rotated", "cuda")
    layer, layer_cache, kwargs = smoothing_inputs(model)
    config = quant_config(None, None)
    config.opts = None
    config.ipts = LlmActivationQuantizerConfig(
        dtype=QDType.sint8, zero_point=None, group_shapes=((-1, -1),), scale_dtypes=(torch.float16,), static=True,
        calib_range=SkipBasedDynamicRangeCalibConfig(
            degree=2, objective=SearchBasedCalibObjective.OutputsError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Layer, max_shrink=0.4, max_expand=1.0, num_grids=4, skips=[]))
    ranges: dict = {}
    hooked = quantize_llm_layer_activations(layer, config, ranges, layer_cache=layer_cache, layer_kwargs=kwargs)
    assert sorted(ranges) == LS.meta["static"]["ranges"] == sorted(hooked)
    for k, state in ranges.items():
        (rng,) = state["dynamic_range"]
        np.testing.assert_array_equal(rng["max"].float().cpu().numpy().reshape(-1), LS[f"static|max|{k}"], err_msg=k)
    tokens = torch.from_numpy(LS["tokens"][0]).cuda()
    with torch.no_grad():
        logits = model(tokens).logits.float().cpu().numpy()
    ref = LS["logits|static"]
    assert np.abs(logits - ref).max() <= 0.03 * np.abs(ref).max()
