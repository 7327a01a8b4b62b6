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


GC = load_golden("gptq_conv")


def conv_columns(w, view):
    """(#g0, gs0, #g1, gs1, #g2, gs2, ...) -> rows x columns with each group's columns side by side
    (deepcompressor/quantizer/kernel/gptq.py:168-178), and the function that undoes it (:272-281)."""
    n = len(view)
    order = [0, 1] + list(range(2, n, 2)) + list(range(3, n, 2))
    mat = w.reshape(view).transpose(order).reshape(view[0] * view[1], -1)
    inverse = np.argsort(order)
    moved = [view[i] for i in order]
    return mat, (lambda q: q.reshape(moved).transpose(inverse).reshape(view))


def unfold3x3(x):
    """[batch][C][H][W] -> [batch * H * W][C * 9], padding 1: the rows a 3 x 3 convolution's weight matrix multiplies."""
    b, c, h, w = x.shape
    padded = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    cols = np.stack([padded[:, :, i:i + h, j:j + w] for i in range(3) for j in range(3)], axis=2)  # b, c, 9, h, w
    return cols.reshape(b, c * 9, h * w).transpose(0, 2, 1).reshape(-1, c * 9)


@pytest.mark.parametrize("tag", sorted(GC.meta))
def test_gptq_oracle_on_a_convolution_weight(tag):
    m = GC.meta[tag]
    spec = qo.spec_from_case(m["case"])
    w, x = GC[f"{tag}|w"], GC[f"{tag}|x"]
    view = m["view_shape"]
    base = qo.fake_quantize(w, "bf16", spec)
    np.testing.assert_array_equal(base["scale"].reshape(-1), GC[f"{tag}|scale"].reshape(-1))
    mat, restore = conv_columns(w, view)
    n_cg, cg = int(np.prod(view[2::2])), int(np.prod(view[3::2]))
    codes = go.gptq_codes(mat, unfold3x3(x), m["num_samples"], spec.steps[0], base["scale"].reshape(view[0], n_cg),
                          base["zero"].reshape(view[0], n_cg), (view[0], view[1], n_cg, cg))
    codes = restore(codes).reshape(w.shape)
    want = GC[f"{tag}|codes"].reshape(w.shape)
    assert (codes == want).mean() > 0.99, float((codes == want).mean())
