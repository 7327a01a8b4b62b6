"""Floating-point types with a zero point (range centred on the group mean): the CPU oracle against reference
fixtures.  The mean is a float32 reduction whose order is not pinned, so a group's scale / zero can sit one unit in
the last place away; groups whose scale and zero agree must agree in every code."""

import numpy as np
import pytest
from conftest import load_golden

import quant_oracle as qo

FZ = load_golden("fp_zero")


def agreement(got_scale, got_zero, got_codes, got_deq, tag, group):
    """(fraction of groups with identical scale and zero, codes / dequant identical inside those groups)."""
    want_scale, want_zero = FZ[f"{tag}|scale"].reshape(-1), FZ[f"{tag}|zero"].reshape(-1)
    same = (np.asarray(got_scale).reshape(-1) == want_scale) & (np.broadcast_to(got_zero, FZ[f"{tag}|zero"].shape)
                                                                 .reshape(-1) == want_zero)
    codes_ok = (np.asarray(got_codes).reshape(-1, group) == FZ[f"{tag}|codes"].reshape(-1, group)).all(axis=1)
    deq_ok = (np.asarray(got_deq).reshape(-1, group) == FZ[f"{tag}|dequant"].reshape(-1, group)).all(axis=1)
    return same.mean(), bool(codes_ok[same].all() and deq_ok[same].all())


@pytest.mark.parametrize("tag", sorted(FZ.meta["entries"]))
def test_mean_centred_range_matches_reference(tag):
    m = FZ.meta["entries"][tag]
    case = FZ.meta["cases"][m["case"]]
    spec = qo.spec_from_case(case)
    dyn = None if m["ratio"] is None else [{"ratio": m["ratio"]}]
    out = qo.fake_quantize(FZ[f"{tag}|x"], m["in_dtype"], spec, dynamic_range=dyn)
    group = case["groups"][-1][1] if case["groups"][-1][1] > 0 else case["shape"][1]
    frac, inside_ok = agreement(out["scale"], out["zero"], out["codes"], out["dequant"], tag, group)
    assert frac >= 0.97 and inside_ok, frac
