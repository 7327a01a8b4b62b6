"""The configurations of the headline workloads at their full sizes, checked through properties that do not depend
on the size -- and, because rows of a weight are independent in fake-quantization, range search and GPTQ, through the
CPU oracle on a slab of rows taken out of the full-size result."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import build_quantizer, to_np
from conftest import assert_bit_equal, load_golden

import gptq_oracle as go
import quant_oracle as qo
from deepcompressor_amd.calib import (DynamicRangeCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                      SearchBasedCalibStrategy, calibrate_dynamic_range)
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.common import TensorType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.quantizer import QuantGptqConfig

pytestmark = pytest.mark.gpu
FQ = load_golden("fake_quant")
SEARCH = DynamicRangeCalibConfig(degree=2.4, objective=SearchBasedCalibObjective.TensorError,
                                 strategy=SearchBasedCalibStrategy.GridSearch,
                                 granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0,
                                 num_grids=100)


def heavy_tailed(rows, cols, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(rows, cols, generator=g) * 0.02
    return torch.where(torch.rand(rows, cols, generator=g) < 0.001, w * 5, w).to(dtype)


def search(q, w):
    return calibrate_dynamic_range(TensorType.Weights, SEARCH, True, q, modules=[None], activations=None,
                                   weights=[nn.Parameter(w, requires_grad=False)])[0]


@pytest.mark.parametrize("cname,dt", [("sint4_g128_f16s", torch.bfloat16), ("uint4_post_g128_f16s", torch.float16)])
def test_q_proj_fake_quant_and_range_search(cname, dt):
    """4096 x 4096 (Llama q_proj), W4 g128 with the 101-candidate search of the GPTQ-R recipe."""
    case = FQ.meta["cases"][cname]
    name = {torch.bfloat16: "bf16", torch.float16: "f16"}[dt]
    w = heavy_tailed(4096, 4096, 0, dt)
    wd = w.cuda()
    q = build_quantizer(case)
    plain = q.quantize(wd, return_with_dequant=False, return_with_quant=True)
    codes, scale = plain.qdata, plain.scale.data
    lo, hi = (-7, 7) if case["zero"] is None else (0, 15)
    assert codes.min() >= lo and codes.max() <= hi and torch.equal(codes, codes.round())
    dq = q.quantize(wd).data
    # dequantized values are exactly code * scale (- zero), rounded once to the weight's dtype
    s_full = scale.expand(4096, 1, 32, 1).reshape(4096, 32, 1)
    z = plain.zero.expand(4096, 1, 32, 1).reshape(4096, 32, 1) if case["zero"] else 0
    assert torch.equal(dq, (codes.view(4096, 32, 128) * s_full - z).to(dt).view(4096, 4096))
    # idempotent: the grid point nearest to a grid point is itself, and the group maximum keeps the scale
    if case["zero"] is None:
        assert torch.equal(q.quantize(dq).data, dq)
    # rows are independent: a slab quantized on its own equals the slab of the full result
    assert torch.equal(q.quantize(wd[1000:1064].contiguous()).data, dq[1000:1064])
    # range search: never worse than the unshrunk range, ranges inside the measured range, rows independent
    rng = search(q, wd)
    spec = qo.spec_from_case(case)
    base_max = wd.float().view(4096, 32, 128).abs().amax(-1) if case["zero"] is None else wd.float().view(4096, 32, 128).amax(-1)
    assert (rng.max.view(4096, 32) <= base_max).all() and (rng.max.view(4096, 32) >= 0.2 * base_max - 1e-12).all()
    slab = slice(2048, 2048 + 48)
    slab_rng = search(build_quantizer(case), wd[slab].contiguous())
    assert torch.equal(slab_rng.max, rng.max[slab])
    # ... and the slab against the CPU oracle, bit for bit: ranges and the dequantized weights they give
    w_np = w[slab].float().numpy()
    best = qo.range_search_group(w_np, name, spec, qo.range_ratios(100), 2.4)
    assert_bit_equal(to_np(rng.max[slab]).reshape(-1), best["max"].reshape(-1), "searched max")
    want = qo.fake_quantize(w_np, name, spec, dynamic_range=[{"max": best["max"], "min": best["min"]}])
    searched = q.quantize(wd, dynamic_range=rng).data
    assert_bit_equal(to_np(searched[slab]), want["dequant"], "dequantized slab")
    err_plain = (dq.float() - wd.float()).abs().pow(2.4).sum()
    err_searched = (searched.float() - wd.float()).abs().pow(2.4).sum()
    assert err_searched <= err_plain


def test_gptq_on_llama_shapes():
    """GPTQ W4 g128 on q_proj (4096 x 4096) and a slab of down_proj (256 x 14336) with 8192 tokens: codes in range,
    better layer output than round-to-nearest, rows independent, and a 16-row slab against the CPU oracle."""
    case = FQ.meta["cases"]["uint4_post_g128_f16s"]
    spec = qo.spec_from_case(case)
    for rows, cols, seed in ((4096, 4096, 1), (256, 14336, 2)):
        g = torch.Generator().manual_seed(seed)
        w = heavy_tailed(rows, cols, seed)
        x = (torch.randn(8192, cols, generator=g) * torch.exp(0.8 * torch.randn(cols, generator=g))).to(torch.bfloat16)
        wd, xd = w.cuda(), x.cuda()
        cache = TensorCache([xd], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=8192, orig_device=xd.device)
        q = build_quantizer(case)
        kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
        out = q.quantize(wd, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True)
        codes = out.qdata
        assert codes.min() >= 0 and codes.max() <= 15 and torch.equal(codes, codes.round())
        dq = q.quantize(wd, kernel=kern, inputs=cache).data
        rtn = q.quantize(wd).data
        sample = xd[:2048].float()
        e_gptq = (sample @ (dq.float() - wd.float()).T).square().sum()
        e_rtn = (sample @ (rtn.float() - wd.float()).T).square().sum()
        assert e_gptq < 0.8 * e_rtn, (float(e_gptq), float(e_rtn))
        slab = slice(rows // 2, rows // 2 + 16)
        again = q.quantize(wd[slab].contiguous(), kernel=kern, inputs=cache, return_with_dequant=False,
                           return_with_quant=True).qdata
        assert torch.equal(again, codes[slab])
        if cols == 4096:  # the oracle factors the 4096^2 Hessian on the CPU in a few seconds
            w_np = w[slab].float().numpy()
            base = qo.fake_quantize(w_np, "bf16", spec)
            want = go.gptq_codes(w_np, x.float().numpy(), 8192, spec.steps[0], base["scale"], base["zero"],
                                 base["view_shape"])
            assert (to_np(codes[slab]) == want).mean() > 0.99


def test_qoq_progressive_feeds_the_qserve_converter_at_full_size():
    """QoQ W4 (sint8 per channel -> uint4 g128 with int8 scales and zeros) on 4096 x 4096 float16: the reference's
    QServe converter arithmetic must find integer first-level weights in [-128, 127] and 4-bit codes in [0, 15]
    (its own assertions), and unpacking its packed bytes must give back the quantizer's codes."""
    from deepcompressor_amd.backend.qserve import convert_to_qserve_w4x8y16_linear_weight

    import qserve_oracle as qs

    case = FQ.meta["cases"]["qoq_g128"]
    w = heavy_tailed(4096, 4096, 3, torch.float16).cuda()
    q = build_quantizer(case)
    out = q.quantize(w, return_with_dequant=False, return_with_quant=True)
    dq = q.quantize(w).data
    leaves = out.scale.state_dict("s")
    packed, s1, zeros, s2 = convert_to_qserve_w4x8y16_linear_weight(dq, leaves["s.0"], out.zero, leaves["s.1"],
                                                                    zero_pre_scaled=False)
    assert packed.shape == (4096, 2048) and s2.shape == (32, 4096) and zeros.dtype == torch.int8
    want = qs.pack_weight(to_np(out.qdata).astype(np.uint8))
    np.testing.assert_array_equal(packed.cpu().numpy(), want)


def test_hiera_w8a8_smoothing_search():
    """SAM2 Hiera-L stage-4 MLP input (1152 channels, fc1 1152 -> 4608), W8A8 per channel / per token with the
    per-channel smoothing search: the chosen scale is finite and positive, never worse than the first candidate, and
    applying it (weights * s, inputs / s) leaves the layer's function unchanged."""
    from deepcompressor_amd.calib.config import SmoothCalibConfig, SmoothSpanMode
    from deepcompressor_amd.calib.smooth import SmoothLinearCalibrator, smooth_linear_modules
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

    g = torch.Generator().manual_seed(4)
    fc1 = nn.Linear(1152, 4608, bias=True).cuda()
    fc1.weight.data = (torch.randn(4608, 1152, generator=g) * 0.03).cuda()
    gains = torch.exp(torch.randn(1152, generator=g) * 1.2)
    xs = [(torch.randn(1, 1024, 1152, generator=g) * gains).cuda() for _ in range(4)]
    cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=4, num_total=4,
                                     num_samples=4, orig_device=torch.device("cuda")))
    wq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(None,)), key="w")
    xq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(None,)),
                   channels_dim=-1, key="x")
    cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=20,
                            strategy=SearchBasedCalibStrategy.GridSearch, objective=SearchBasedCalibObjective.OutputsError,
                            granularity=SearchBasedCalibGranularity.Layer)
    w0 = fc1.weight.data.clone()
    calib = SmoothLinearCalibrator(cfg, wq, xq)
    best = calib.calibrate(x_wgts=[fc1.weight], x_acts=cache, x_mods=[fc1], eval_inputs=cache, eval_module=fc1,
                           eval_kwargs={})
    assert torch.equal(fc1.weight.data, w0)
    errors = np.array([float(e[0]) for e in calib.error_history])
    assert len(errors) == 39 and np.isfinite(errors).all() and errors.min() <= errors[0]
    assert best.shape == (1152,) and torch.isfinite(best).all() and (best > 0).all()
    with torch.inference_mode():
        y_ref = fc1(xs[0])
        smooth_linear_modules(None, [fc1], scale=best.cuda(), config=cfg, weight_quantizer=wq, input_quantizer=xq,
                              inputs=cache, eval_inputs=cache, eval_module=fc1)
        torch.testing.assert_close(fc1(xs[0] / best.cuda()), y_ref, rtol=2e-4, atol=2e-4)
