"""The CPU GPTQ oracle against the fixtures captured from the reference (tolerance-based, see gptq_oracle.py)."""

import numpy as np
import pytest
from conftest import load_golden

import gptq_oracle as go
import quant_oracle as qo

GQ = load_golden("gptq")
FQ = load_golden("fake_quant")


def layer_output_error(x, w_a, w_b):
    """max |x @ (w_a - w_b)^T| / max |x @ w_b^T|  in float64."""
    x = x.reshape(-1, x.shape[-1]).astype(np.float64)
    ya, yb = x @ w_a.astype(np.float64).T, x @ w_b.astype(np.float64).T
    return np.abs(ya - yb).max() / np.abs(yb).max()


@pytest.mark.parametrize("tag", sorted(GQ.meta))
def test_gptq_oracle_matches_reference(tag):
    m = GQ.meta[tag]
    case = FQ.meta["cases"][m["case"]]
    spec = qo.spec_from_case(case)
    w, x = GQ[f"{tag}|w"], GQ[f"{tag}|x"]
    x2 = x.reshape(-1, x.shape[-1])
    base = qo.fake_quantize(w, m["in_dtype"], spec)  # scale / zero come from the un-updated weights
    np.testing.assert_array_equal(base["scale"], GQ[f"{tag}|scale"])
    codes = go.gptq_codes(w, x2, m["num_samples"], spec.steps[0], base["scale"], base["zero"], base["view_shape"])
    want = GQ[f"{tag}|codes"]
    same = (codes == want).mean()
    assert same > 0.99, f"only {same:.4f} of the codes equal the reference's"
    # dequantize both with the same scale/zero and compare on the layer output
    step = spec.steps[0]
    deq = lambda q: dequant(q, base, step, w.shape)  # noqa: E731
    err = layer_output_error(x, deq(codes), deq(want))
    assert err < 1e-3, f"layer output differs by {err:.2e}"
    # and GPTQ actually helps: output error vs the fp weights is below round-to-nearest's
    e_gptq = layer_output_error(x, deq(codes), w) 
    e_rtn = layer_output_error(x, GQ[f"{tag}|rtn_dequant"], w)
    assert e_gptq < e_rtn


def dequant(q, base, step, shape):
    view = base["view_shape"]
    d = q.reshape(view).astype(np.float32)
    if step.zero_domain == "PreScale":
        d = d - base["zero"]
    d = d * base["scale"]
    if step.zero_domain == "PostScale":
        d = d - base["zero"]
    return d.reshape(shape).astype(np.float32)
