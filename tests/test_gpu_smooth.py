"""Smooth quantization on the GPU against the reference's fixtures and the CPU oracle.

Bars: per-channel abs-max, applied scales (weights, biases, grouped-query layout): bit-exact.  Per-channel means /
RMS: float32 sums in a different order, <= 4 ulp.  Smoothing scales: <= 2 ulp (the reference's float32 ``pow`` is
accurate to 1 ulp, not correctly rounded).  Search: every candidate's error within 2e-3 relative (GEMM summation
order) and the same candidate chosen.
"""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import to_device, to_np
from conftest import assert_bit_equal, load_golden

import smooth_oracle as so
from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                             SearchBasedCalibStrategy, SmoothCalibConfig, SmoothSpanMode)
from deepcompressor_amd.calib.metric import ChannelMetric
from deepcompressor_amd.calib.smooth import (ActivationSmoother, SmoothLinearCalibrator, get_smooth_scale,
                                             smooth_attention, smooth_linear_modules)
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

pytestmark = pytest.mark.gpu
G = load_golden("smooth")
TORCH = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def _ulps(got, want):
    return np.abs(got.astype(np.float64) - want) / np.spacing(np.abs(want).astype(np.float32))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_channel_metric_fixture(dt):
    xs = [to_device(G[f"span|{dt}|x{i}"], dt) for i in range(3)]
    ws = [to_device(G[f"span|{dt}|w{i}"], dt) for i in range(2)]
    conv = to_device(G[f"span|{dt}|conv"], dt)
    for name, tensors in (("x", xs), ("w", ws), ("conv", conv)):
        got = ChannelMetric.abs_max(tensors, 96, [1, -1])
        assert got.dtype == torch.float32 and got.shape == (96,)
        assert_bit_equal(to_np(got), G[f"span|{dt}|{name}|abs_max"], f"{name} abs_max")
        for mode in ("abs_mean", "root_mean_square"):
            got = to_np(getattr(ChannelMetric, mode)(tensors, 96, [1, -1]))
            assert _ulps(got, G[f"span|{dt}|{name}|{mode}"]).max() <= 4, (name, mode)


def test_channel_metric_large_vs_oracle():
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(5000, 1000, generator=g) * torch.exp(torch.randn(1000, generator=g))).to(torch.bfloat16)
    x[17, 3] = float("nan")
    xn = x.float().numpy()
    got = to_np(ChannelMetric.abs_max([x.cuda()], 1000, [1, -1]))
    want = so.channel_metric([xn], 1000, "abs_max")
    assert np.isnan(got[3]) and np.isnan(want[3])  # amax propagates NaN
    assert np.array_equal(np.delete(got, 3), np.delete(want, 3))
    rms = to_np(ChannelMetric.root_mean_square([x.cuda()[:, 4:]], 996, [1, -1]))
    assert _ulps(rms, so.channel_metric([xn[:, 4:]], 996, "root_mean_square")).max() <= 4


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_smooth_scale_fixture(dt):
    a = torch.from_numpy(G[f"span|{dt}|x|abs_max"]).cuda()
    b = torch.from_numpy(G[f"span|{dt}|w|abs_max"]).cuda()
    for alpha, beta in ((0.5, 0.5), (0.3, 0.7), (0.85, 0), (0, 0.4), (1.0, 1.0)):
        got = to_np(get_smooth_scale(alpha_base=a.clone(), beta_base=b.clone(), alpha=alpha, beta=beta))
        want = G[f"scale|{dt}|{alpha}|{beta}"]
        assert _ulps(got, want).max() <= 2, (alpha, beta)
        assert (got == 1).sum() == (want == 1).sum()  # the dead channel maps to 1, not 0


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_apply_scale_fixture(dt):
    prev = nn.Linear(48, 64, bias=True, device="cuda", dtype=TORCH[dt])
    mods = [nn.Linear(64, 40, bias=False, device="cuda", dtype=TORCH[dt]),
            nn.Linear(64, 24, bias=False, device="cuda", dtype=TORCH[dt])]
    prev.weight.data, prev.bias.data = to_device(G[f"apply|{dt}|prev_w"], dt), to_device(G[f"apply|{dt}|prev_b"], dt)
    for i, m in enumerate(mods):
        m.weight.data = to_device(G[f"apply|{dt}|w{i}"], dt)
    out = smooth_linear_modules(prev, mods, scale=torch.from_numpy(G[f"apply|{dt}|scale"]))
    assert out.device.type == "cpu"
    assert_bit_equal(to_np(prev.weight.data), G[f"apply|{dt}|prev_w_out"], "prev w")
    assert_bit_equal(to_np(prev.bias.data), G[f"apply|{dt}|prev_b_out"], "prev b")
    for i, m in enumerate(mods):
        assert_bit_equal(to_np(m.weight.data), G[f"apply|{dt}|w{i}_out"], f"w{i}")
    # grouped-query layout
    prev2 = nn.Linear(48, 32, bias=False, device="cuda", dtype=TORCH[dt])
    mod2 = nn.Linear(64, 40, bias=False, device="cuda", dtype=TORCH[dt])
    prev2.weight.data, mod2.weight.data = to_device(G[f"apply_gqa|{dt}|prev_w"], dt), to_device(G[f"apply_gqa|{dt}|w"], dt)
    smooth_linear_modules(prev2, mod2, scale=torch.from_numpy(G[f"apply_gqa|{dt}|scale"]), num_heads=4, num_head_repeats=2)
    assert_bit_equal(to_np(prev2.weight.data), G[f"apply_gqa|{dt}|prev_w_out"], "gqa prev")
    assert_bit_equal(to_np(mod2.weight.data), G[f"apply_gqa|{dt}|w_out"], "gqa w")
    # query / key smoothing with a given scale: q rows up, k rows down; the activation-side divider undoes an upscale
    q, k = nn.Linear(16, 64, bias=False, device="cuda", dtype=TORCH[dt]), nn.Linear(16, 32, bias=False, device="cuda", dtype=TORCH[dt])
    q0, k0 = q.weight.data.clone(), k.weight.data.clone()
    s = torch.from_numpy(G[f"apply_gqa|{dt}|scale"])
    smooth_attention(q_proj=q, k_proj=k, scale=s, num_heads=4, num_head_repeats=2)
    assert_bit_equal(to_np(q.weight.data), so.upscale(to_np(q0), dt, s.numpy(), 0), "q")
    assert_bit_equal(to_np(k.weight.data), so.downscale(to_np(k0), dt, so.downscale_for_producer(s.numpy(), 4, 2), 0), "k")
    x = torch.randn(7, 64, device="cuda").to(TORCH[dt])
    got = ActivationSmoother(s, channels_dim=-1, develop_dtype=torch.float32).process(x)
    assert_bit_equal(to_np(got), so.downscale(to_np(x), dt, s.numpy(), 1), "activation smoother")


class Pair(nn.Module):
    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b

    def forward(self, x):
        return torch.cat([self.a(x), self.b(x)], dim=-1)


@pytest.mark.parametrize("tag", ["grid", "manual"])
def test_search_fixture(tag):
    m = G.meta[f"search|{tag}"]
    a, b = nn.Linear(64, 40, bias=False, device="cuda"), nn.Linear(64, 24, bias=False, device="cuda")
    a.weight.data, b.weight.data = torch.from_numpy(G[f"search|{tag}|wa"]).cuda(), torch.from_numpy(G[f"search|{tag}|wb"]).cuda()
    wa0 = a.weight.data.clone()
    xs = [torch.from_numpy(G[f"search|{tag}|x{i}"]).cuda() for i in range(3)]
    wq = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 16),), scale_dtypes=(torch.float16,)), key="w")
    xq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(torch.float16,)),
                   channels_dim=-1, key="x")
    cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=m["alpha"], beta=m["beta"],
                            num_grids=20, strategy=SearchBasedCalibStrategy[m["strategy"]],
                            objective=SearchBasedCalibObjective.OutputsError, granularity=SearchBasedCalibGranularity.Layer)
    cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3, num_samples=6,
                                     orig_device=torch.device("cuda")))
    calib = SmoothLinearCalibrator(cfg, wq, xq)
    best = calib.calibrate(x_wgts=[a.weight, b.weight], x_acts=cache, x_mods=[a, b], eval_inputs=cache,
                           eval_module=Pair(a, b), eval_kwargs={})
    assert torch.equal(a.weight.data, wa0), "the search must leave the weights as they were"
    errors = np.array([float(e[0]) for e in calib.error_history])
    want = G[f"search|{tag}|errors"]
    assert errors.shape == want.shape
    assert np.all(np.abs(errors - want) <= 2e-3 * want), np.abs(errors / want - 1).max()
    assert int(np.argmin(errors)) == m["best_index"]
    assert _ulps(to_np(best), G[f"search|{tag}|best_scale"]).max() <= 2
