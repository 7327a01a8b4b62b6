"""Activation range search on the GPU (calibrate_dynamic_range with TensorType.Inputs / Outputs) against the
reference fixtures and the CPU oracle."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import TORCH_DT, build_quantizer, to_device, to_np
from _toy import ToyGated
from conftest import load_golden
from test_oracle_range_acts import ENTRIES, RA, check, run_oracle, tolerance

from deepcompressor_amd.calib import (DynamicRangeCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                      SearchBasedCalibStrategy, calibrate_dynamic_range)
from deepcompressor_amd.calib.range import DynamicRangeCalibrator
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.common import TensorType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn

pytestmark = pytest.mark.gpu


def setup(tag):
    m = ENTRIES[tag]
    dt = m["in_dtype"]
    mod = ToyGated(m["channels"], m["hidden"]).to(TORCH_DT[dt]).cuda()
    for name in ("up", "gate", "down"):
        getattr(mod, name).weight = nn.Parameter(to_device(RA[f"{tag}|w_{name}"], dt), requires_grad=False)
    xs = [to_device(RA[f"{tag}|x{i}"], dt) for i in range(3)]
    cache = lambda data: TensorsCache(TensorCache([d.clone() for d in data], channels_dim=-1,  # noqa: E731
                                                  reshape=LinearReshapeFn(), num_cached=3, num_total=3, num_samples=6,
                                                  orig_device=xs[0].device))
    q = build_quantizer(RA.meta["cases"][m["case"]], channels_dim=-1)
    cfg = DynamicRangeCalibConfig(
        degree=m["degree"], objective=SearchBasedCalibObjective[m["objective"]],
        strategy=SearchBasedCalibStrategy[m["strategy"]], granularity=SearchBasedCalibGranularity[m["granularity"]],
        max_shrink=0.4, max_expand=1.0, num_grids=m["num_grids"], allow_scale=m["allow_scale"],
        element_batch_size=-1, element_size=-1, sample_batch_size=-1, sample_size=-1)
    if m["tensor_type"] == "Inputs":
        modules, acts = [mod.up, mod.gate], cache(xs)
    else:
        with torch.inference_mode():
            modules, acts = [mod.mid], cache([mod.up(x) * torch.sigmoid(mod.gate(x)) for x in xs])
    return m, mod, xs, cache, q, cfg, modules, acts


@pytest.mark.parametrize("tag", sorted(t for t, m in ENTRIES.items() if m["reference_raised"] is None))
def test_activation_search_matches_reference_and_oracle(tag):
    m, mod, xs, cache, q, cfg, modules, acts = setup(tag)
    seen = {}
    orig = DynamicRangeCalibrator.calibrate

    def spy(self, *a, **kw):
        seen["calib"] = self
        return orig(self, *a, record_errors=True, **kw)

    DynamicRangeCalibrator.calibrate = spy
    try:
        ranges = calibrate_dynamic_range(TensorType[m["tensor_type"]], cfg, m["static"], q, modules, acts,
                                         eval_inputs=cache(xs), eval_module=mod, eval_kwargs={})
    finally:
        DynamicRangeCalibrator.calibrate = orig
    calib = seen["calib"]
    assert calib.objective.name == m["resolved_objective"] and calib.granularity.name == m["resolved_granularity"]
    (rng,) = ranges
    got = dict(errors=list(to_np(calib.candidate_errors)), ratio=to_np(calib.best_range.ratio),
               max=None if rng.max is None else to_np(rng.max), min=None if rng.min is None else to_np(rng.min))
    check(m, tag, got, rtol_err=tolerance(m))
    # the shape and kind of what is handed back (calib/range.py:152-164)
    if m["static"]:
        assert rng.ratio is None and tuple(rng.max.shape) == RA[f"{tag}|s0|max"].shape
    elif not m["allow_scale"]:
        assert rng.ratio == 1.0 and rng.max is not None
    else:
        assert rng.max is None and isinstance(rng.ratio, torch.Tensor)
        assert tuple(rng.ratio.shape) == RA[f"{tag}|s0|out_ratio"].shape
        np.testing.assert_array_equal(to_np(rng.ratio), RA[f"{tag}|s0|out_ratio"])
    # against the oracle on the same inputs (TensorError: same kernels' float32 group sums, so the errors agree to
    # the unit in the last place that the batch sums allow)
    _, want = run_oracle(tag)
    np.testing.assert_allclose(np.stack(got["errors"]).reshape(-1), np.stack(want["errors"]).reshape(-1),
                               rtol=tolerance(m))
    # the result feeds the quantizer: hooks with the calibrated range run and change the module's output
    q.dynamic_range = rng
    x = xs[0]
    with torch.inference_mode():
        y0 = mod(x)
        hook = q.as_hook(is_output=m["tensor_type"] == "Outputs").register(modules)
        y1 = mod(x)
        hook.remove()
    assert y1.shape == y0.shape and not torch.equal(y0, y1) and torch.isfinite(y1).all()


@pytest.mark.parametrize("tag", sorted(t for t, m in ENTRIES.items() if m["reference_raised"] is not None))
def test_group_granularity_on_inputs_raises_like_the_reference(tag):
    m, mod, xs, cache, q, cfg, modules, acts = setup(tag)
    with pytest.raises(ValueError, match="Unknown granularity"):
        calibrate_dynamic_range(TensorType.Inputs, cfg, True, q, modules, acts, eval_inputs=cache(xs), eval_module=mod)


def test_static_output_range_is_the_measured_range():
    """smoothquant-static style: per-tensor int8 outputs, Manual strategy -> the stored range is max |y| over the cache."""
    tag = next(t for t, m in ENTRIES.items() if m["strategy"] == "Manual")
    m, mod, xs, cache, q, cfg, modules, acts = setup(tag)
    (rng,) = calibrate_dynamic_range(TensorType.Outputs, cfg, True, q, modules, acts, eval_inputs=cache(xs),
                                     eval_module=mod, eval_kwargs={})
    want = max(float(t.abs().max()) for t in acts.front().data)
    assert float(rng.max) == want and rng.min is None
    sd = q.state_dict()
    assert set(sd) >= {"channels_dim", "scale", "zero", "dynamic_range", "range_bound", "quant_range"}
