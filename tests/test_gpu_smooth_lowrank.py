"""Smoothing search scored with the low-rank branch in place (allow_low_rank) on the GPU against reference
fixtures: HIP channel scaling / fake quantization through the C ABI, float64 SVD through torch."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import to_np
from test_oracle_smooth_lowrank import G, config

from deepcompressor_amd.calib.smooth import SmoothLinearCalibrator
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig
from deepcompressor_amd.quantizer.config.lowrank import QuantLowRankConfig

pytestmark = pytest.mark.gpu


class Pair(nn.Module):
    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b

    def forward(self, x):
        return torch.cat([self.a(x), self.b(x)], dim=-1)


@pytest.mark.parametrize("tag", sorted(G.meta))
def test_low_rank_aware_search_fixture(tag):
    m = G.meta[tag]
    a, b = nn.Linear(64, 40, bias=False, device="cuda"), nn.Linear(64, 24, bias=False, device="cuda")
    a.weight.data, b.weight.data = torch.from_numpy(G[f"{tag}|wa"]).cuda(), torch.from_numpy(G[f"{tag}|wb"]).cuda()
    wa0, wb0 = a.weight.data.clone(), b.weight.data.clone()
    xs = [torch.from_numpy(G[f"{tag}|x{i}"]).cuda() for i in range(3)]
    wq = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 16),), scale_dtypes=(torch.float16,)),
                   key="w", low_rank=QuantLowRankConfig(**m["low_rank"]))
    xq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(torch.float16,)),
                   channels_dim=-1, key="x")
    cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3, num_samples=6,
                                     orig_device=torch.device("cuda")))
    calib = SmoothLinearCalibrator(config(), wq, xq)
    pair = Pair(a, b)
    y_before = pair(xs[0]).clone()
    best = calib.calibrate(x_wgts=[a.weight, b.weight], x_acts=cache, x_mods=[a, b], eval_inputs=cache,
                           eval_module=pair, eval_kwargs={}, splits=m["splits"])
    assert torch.equal(a.weight.data, wa0) and torch.equal(b.weight.data, wb0), "weights must be restored"
    assert torch.equal(pair(xs[0]), y_before), "every hook must be removed"
    errors = np.array([float(e[0]) for e in calib.error_history])
    np.testing.assert_allclose(errors, G[f"{tag}|errors"], rtol=1e-2)
    assert int(np.argmin(errors)) == m["best_index"]
    np.testing.assert_allclose(to_np(best), G[f"{tag}|best_scale"], rtol=1e-6)
