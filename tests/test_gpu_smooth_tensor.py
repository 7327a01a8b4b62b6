"""Smoothing search scored on the weights alone (TensorError) on the GPU against the reference's fixtures.

Bars: candidate scales <= 4 ulp (a quotient of two float32 powers, each within 1 ulp of the reference's); every
candidate's errors within 1e-4 relative for float32 weights (sums of 640+ terms in another order, on weights processed
with scales that differ in the last bits) and within two units of the last place of float16 / bfloat16 errors; the
final scale equal to what the reference's selection rule gives on the errors measured here (exact), and equal to the
reference's final scale in every group whose decision was not a near tie."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import to_device, to_np
from test_oracle_smooth_tensor import ST, case_inputs

import smooth_oracle as so
from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                             SearchBasedCalibStrategy, SmoothCalibConfig, SmoothSpanMode)
from deepcompressor_amd.calib.smooth import SmoothLinearCalibrator
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

pytestmark = pytest.mark.gpu
TORCH = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
TOL = {"f32": 1e-4, "f16": 2.0 ** -9, "bf16": 2.0 ** -6}


@pytest.mark.parametrize("tag", sorted(ST.meta))
def test_tensor_error_search_matches_reference(tag):
    m, weights, xs = case_inputs(tag)
    dt = m["dtype"]
    mods = [nn.Linear(64, r, bias=False, device="cuda", dtype=TORCH[dt]) for r in m["rows"]]
    for mod, w in zip(mods, weights):
        mod.weight.data = to_device(w, dt)
    xs = [to_device(x, dt) for x in xs]
    wq = Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 16),),
                                   scale_dtypes=(torch.float16,)), key="w")
    cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=10,
                            strategy=SearchBasedCalibStrategy.GridSearch,
                            objective=SearchBasedCalibObjective[m["objective"]],
                            granularity=SearchBasedCalibGranularity[m["granularity"]], degree=2)
    cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3,
                                     num_samples=6, orig_device=torch.device("cuda")))
    calib = SmoothLinearCalibrator(cfg, wq, None, num_heads=m["num_heads"], num_head_repeats=m["num_head_repeats"])
    kept = [mod.weight.data.clone() for mod in mods]
    scales, ask = [], calib.ask

    def spy():
        scales.append(ask().clone())
        return scales[-1]

    calib.ask = spy
    best = calib.calibrate(x_wgts=[mod.weight for mod in mods], x_acts=cache, x_mods=mods, eval_inputs=cache,
                           eval_module=None, eval_kwargs={})
    assert calib.objective == SearchBasedCalibObjective.TensorError
    for mod, w in zip(mods, kept):
        assert torch.equal(mod.weight.data, w), "the search must leave the weights as they were"
    errors = np.stack([np.stack([to_np(e).reshape(-1) for e in cand]) for cand in calib.error_history])
    scales = np.stack([to_np(s) for s in scales])
    want_e, want_s, want_best = ST[f"{tag}|errors"], ST[f"{tag}|scales"], ST[f"{tag}|best_scale"]
    assert errors.shape == want_e.shape and scales.shape == want_s.shape
    assert (np.abs(scales.astype(np.float64) - want_s) <= 4 * np.spacing(want_s)).all()
    assert np.all(np.abs(errors - want_e) <= TOL[dt] * want_e), np.abs(errors / want_e - 1).max()
    best_e = best_s = None
    for c in range(m["num_candidates"]):
        best_e, best_s = so.update_best(best_e, best_s, [errors[c, j].copy() for j in range(errors.shape[1])], scales[c],
                                        64, m["num_heads"], m["num_head_repeats"])
    got = to_np(best)
    assert got.shape == want_best.shape
    np.testing.assert_array_equal(got, best_s.reshape(got.shape))
    rows = lambda a: a.reshape(-1, 16) if m["granularity"] != "Layer" else a.reshape(1, -1)  # noqa: E731
    same = (np.abs(rows(got).astype(np.float64) - rows(want_best)) <= 4 * np.spacing(rows(want_best))).all(axis=1)
    assert same.mean() >= 0.75, same
