"""Weight range search scored on products / outputs (HIP fake-quant + dcq_product_error, through the C ABI) against
fixtures captured from the reference and against the oracle; GEMM summation order makes this a tolerance check."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import build_quantizer, to_device, to_np
from test_oracle_range_products import FQ, RP, check_against_fixture, run_oracle

from deepcompressor_amd.calib import (DynamicRangeCalibConfig, DynamicRangeCalibrator, SearchBasedCalibGranularity,
                                      SearchBasedCalibObjective, SearchBasedCalibStrategy, calibrate_dynamic_range)
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.common import TensorType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn

pytestmark = pytest.mark.gpu


def run_device(tag, monkeypatch):
    m = RP.meta[tag]
    cfg = DynamicRangeCalibConfig(degree=m["degree"], objective=SearchBasedCalibObjective[m["objective"]],
                                  strategy=SearchBasedCalibStrategy.GridSearch,
                                  granularity=SearchBasedCalibGranularity[m["granularity"]], max_shrink=0.8,
                                  max_expand=1.0, num_grids=m["num_grids"],
                                  element_batch_size=m["element_batch_size"], element_size=-1, sample_batch_size=-1,
                                  sample_size=-1)
    calibrators = []
    plain = DynamicRangeCalibrator.calibrate

    def traced(self, *args, **kwargs):
        calibrators.append(self)
        kwargs["record_errors"] = True
        return plain(self, *args, **kwargs)

    monkeypatch.setattr(DynamicRangeCalibrator, "calibrate", traced)
    w = to_device(RP[f"{tag}|w"], m["in_dtype"])
    lin = nn.Linear(w.shape[1], w.shape[0], bias=True, device=w.device, dtype=w.dtype)
    lin.weight = nn.Parameter(w.clone(), requires_grad=False)
    lin.bias = nn.Parameter(to_device(RP[f"{tag}|bias"], m["in_dtype"]), requires_grad=False)
    acts = TensorsCache(TensorCache([to_device(RP[f"{tag}|x{i}"], m["in_dtype"]) for i in range(2)], channels_dim=-1,
                                    reshape=LinearReshapeFn(), num_cached=2, num_total=2, num_samples=4,
                                    orig_device=w.device))
    ranges = calibrate_dynamic_range(TensorType.Weights, cfg, True, build_quantizer(FQ.meta["cases"][m["case"]]),
                                     modules=[lin], activations=acts)
    assert torch.equal(lin.weight.data, w), "the search must leave the weight as it found it"
    return m, ranges, calibrators


@pytest.mark.parametrize("tag", sorted(t for t, m in RP.meta.items() if m["reference_raised"] is None))
def test_products_search_golden(tag, monkeypatch):
    m, ranges, (cal,) = run_device(tag, monkeypatch)
    assert cal.objective.name == m["resolved_objective"] and cal.granularity.name == m["resolved_granularity"]
    check_against_fixture(m, tag, to_np(cal.candidate_errors), to_np(cal.best_range.ratio), to_np(ranges[0].max))


@pytest.mark.parametrize("tag", sorted(t for t, m in RP.meta.items() if m["resolved_objective"] == "ProductsError"))
def test_products_search_against_oracle(tag, monkeypatch):
    """ProductsError, every dtype: the device forms each product as the oracle does -- accumulated in float64 and rounded
    once to float32, then to the dtype -- so the two agree on every candidate's error up to the float32 summation of the
    error terms (1e-5), pick the same ratio wherever the best two candidates are further apart than that, and store
    the same range then; ChannelGroup included (which the reference cannot finish, calib/range.py:208-215)."""
    m, ranges, (cal,) = run_device(tag, monkeypatch)
    _, want = run_oracle(tag)
    got = to_np(cal.candidate_errors)
    want_errors = np.stack(want["errors"]).reshape(got.shape)
    np.testing.assert_allclose(got, want_errors, rtol=1e-5, atol=1e-30)
    srt = np.sort(want_errors.reshape(want_errors.shape[0], -1), axis=0)
    clear = (srt[1] - srt[0]) > 1e-4 * srt[1]
    same = to_np(cal.best_range.ratio).reshape(-1) == want["ratio"].reshape(-1)
    assert same[clear].all() and same.mean() > 0.99, (same.mean(), clear.mean())
    if same.all():
        np.testing.assert_array_equal(to_np(ranges[0].max), want["max"])


@pytest.mark.parametrize("tag", sorted(t for t, m in RP.meta.items()
                                       if m["resolved_objective"] == "OutputsError" and m["in_dtype"] == "f32"))
def test_outputs_search_against_oracle(tag, monkeypatch):
    """OutputsError on float32 modules (the module's own forward produces the outputs): to GEMM rounding."""
    m, ranges, (cal,) = run_device(tag, monkeypatch)
    _, want = run_oracle(tag)
    got = to_np(cal.candidate_errors)
    np.testing.assert_allclose(got, np.stack(want["errors"]).reshape(got.shape), rtol=2e-3, atol=1e-12)
    same = to_np(cal.best_range.ratio).reshape(-1) == want["ratio"].reshape(-1)
    assert same.mean() > 0.97
    if same.all():
        np.testing.assert_array_equal(to_np(ranges[0].max), want["max"])
