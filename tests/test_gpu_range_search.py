"""Fused range-ratio search (HIP, through the C ABI) against reference fixtures and the oracle -- bit-exact."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import build_quantizer, to_device, to_np
from conftest import assert_bit_equal, load_golden

import quant_oracle as qo
from deepcompressor_amd import engine
from deepcompressor_amd.calib import (DynamicRangeCalibConfig, DynamicRangeCalibrator, SearchBasedCalibGranularity,
                                      SearchBasedCalibObjective, SearchBasedCalibStrategy, calibrate_dynamic_range)
from deepcompressor_amd.data.common import TensorType

pytestmark = pytest.mark.gpu
FQ = load_golden("fake_quant")
RS = load_golden("range_search")
TB = load_golden("tables")


def _config(degree, grids):
    return DynamicRangeCalibConfig(degree=degree, objective=SearchBasedCalibObjective.TensorError,
                                   strategy=SearchBasedCalibStrategy.GridSearch,
                                   granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0,
                                   num_grids=grids)


@pytest.mark.parametrize("dt,end", [("bf16", 0x7F80), ("f16", 0x7C00)])
@pytest.mark.parametrize("degree", [2.4, 2, 3, 1.5, 2.5, 4, 1, 2.2])
def test_pow_table_exhaustive(dt, end, degree):
    """All finite non-negative 16-bit inputs: the GPU-built table equals torch-CPU pow_ on that dtype."""
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[dt]
    table = engine.pow_table(torch.device("cuda:0"), dtype, degree).cpu().numpy().view(np.uint16)
    want = TB[f"pow|{dt}|{degree}"]
    bad = table[:end] != want
    assert not bad.any(), f"{int(bad.sum())} entries differ, first {int(np.flatnonzero(bad)[0])}"


@pytest.mark.parametrize("tag", sorted(RS.meta))
def test_range_search_golden(tag):
    m = RS.meta[tag]
    case = FQ.meta["cases"][m["case"]]
    w = to_device(RS[f"{tag}|x"], m["in_dtype"])
    q = build_quantizer(case)
    cal = DynamicRangeCalibrator(TensorType.Weights, _config(m["degree"], m["num_grids"]), static=True, quantizer=q)
    best = cal.calibrate(x_wgts=[nn.Parameter(w, requires_grad=False)], record_errors=True)
    got_err = to_np(cal.candidate_errors)
    want_err = RS[f"{tag}|errors"]
    if m["in_dtype"] == "f32":
        # float32 storage: the reference's vectorised float32 pow is not correctly rounded ("parity unpinned"):
        # errors agree to float32 rounding, the selected ratio on all but near ties
        np.testing.assert_allclose(got_err, want_err, rtol=3e-6)
        agree = (to_np(cal.best_range.ratio).reshape(-1) == RS[f"{tag}|ratio"].reshape(-1)).mean()
        assert agree > 0.98
        return
    for k in range(want_err.shape[0]):
        assert_bit_equal(got_err[k], want_err[k], f"candidate {k} error")
    assert_bit_equal(to_np(cal.best_error), RS[f"{tag}|best_error"], "best error")
    assert_bit_equal(to_np(cal.best_range.ratio), RS[f"{tag}|ratio"], "best ratio")
    assert_bit_equal(to_np(best.max), RS[f"{tag}|max"], "best max")
    if f"{tag}|min" in RS:
        assert_bit_equal(to_np(best.min), RS[f"{tag}|min"], "best min")


@pytest.mark.parametrize("cname,dt", [("sint4_g128_f16s", "bf16"), ("uint4_post_g128_f16s", "bf16"),
                                      ("uint4_pre_g128_f16s", "f16"), ("sint4_g128_f16s", "f16")])
def test_range_search_on_rounding_ties(cname, dt):
    """Weights on a half-unit grid: with the group maximum at the top code, x / scale lands exactly on k + 0.5 for a
    large share of the elements and candidates (round-half-to-even decides the code), and within a few units in the
    last place of it for others -- the cases where an approximate quotient would pick the other code."""
    case = FQ.meta["cases"][cname]
    g = torch.Generator().manual_seed(7)
    top = 7.0 if cname.startswith("sint4") else 7.5
    w = torch.randint(-15, 16, (96, 1024), generator=g).float() * 0.5
    w = w.clamp_(-top, top)
    w[:, ::128] = top      # every group's maximum
    w[:, 1::128] = -top    # and minimum
    w = w * torch.tensor([1.0, 0.5, 3.0, 0.0625, 1.0 / 3.0, 1e-3]).repeat(16).view(96, 1)
    wt = w.to({"bf16": torch.bfloat16, "f16": torch.float16}[dt])
    ratios = qo.range_ratios(100, 0.2, 1.0)
    want = qo.range_search_group(wt.float().numpy(), dt, qo.spec_from_case(case), ratios, 2.4)
    q = build_quantizer(case)
    out = calibrate_dynamic_range(TensorType.Weights, _config(2.4, 100), True, q, modules=[None], activations=None,
                                  weights=[nn.Parameter(wt.cuda(), requires_grad=False)])
    assert_bit_equal(to_np(out[0].max), want["max"], "max")
    if want["min"] is not None:
        assert_bit_equal(to_np(out[0].min), want["min"], "min")


@pytest.mark.parametrize("cname,dt,degree", [("sint4_g128_f16s", "bf16", 2.4), ("uint4_post_g128_f16s", "f16", 2.4),
                                             ("sint4_g64_defs", "bf16", 2), ("uint4_pre_g128_f16s", "bf16", 2.4)])
def test_range_search_vs_oracle_random(cname, dt, degree):
    case = FQ.meta["cases"][cname]
    g = torch.Generator().manual_seed(99)
    w = torch.randn(192, 1024, generator=g) * 0.02
    w = torch.where(torch.rand(192, 1024, generator=g) < 0.003, w * 8, w)
    wt = w.to({"bf16": torch.bfloat16, "f16": torch.float16}[dt])
    ratios = qo.range_ratios(100, 0.2, 1.0)
    want = qo.range_search_group(wt.float().numpy(), dt, qo.spec_from_case(case), ratios, degree)
    q = build_quantizer(case)
    out = calibrate_dynamic_range(TensorType.Weights, _config(degree, 100), True, q, modules=[None], activations=None,
                                  weights=[nn.Parameter(wt.cuda(), requires_grad=False)])
    assert_bit_equal(to_np(out[0].max), want["max"], "max")
    if want["min"] is not None:
        assert_bit_equal(to_np(out[0].min), want["min"], "min")
    # the calibrated static range reproduces through the fused fake-quant path
    deq = q.quantize(wt.cuda(), dynamic_range=out[0])
    ref = qo.fake_quantize(wt.float().numpy(), dt, qo.spec_from_case(case),
                           dynamic_range=[{"max": want["max"], "min": want["min"]}])
    assert_bit_equal(to_np(deq.data), ref["dequant"], "dequant with searched range")
