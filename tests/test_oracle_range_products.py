"""The CPU oracle's products- / outputs-scored weight range search against fixtures captured from the reference.

The reference's GEMM accumulates in float32 in a backend-specific order, so the comparison is by tolerance: tight
for float32 operands, statistical for 16-bit ones (one product landing on the neighbouring 16-bit value moves a
group's error by as much as the quantization effect itself)."""

import numpy as np
import pytest
from conftest import load_golden

import quant_oracle as qo

FQ = load_golden("fake_quant")
RP = load_golden("range_products")


def run_oracle(tag):
    m = RP.meta[tag]
    spec = qo.spec_from_case(FQ.meta["cases"][m["case"]])
    ratios = qo.range_ratios(m["num_grids"], 0.8, 1.0)
    assert np.array_equal(np.asarray(ratios, dtype=np.float64), RP[f"{tag}|ratios"])
    xs = [RP[f"{tag}|x{i}"] for i in range(2)]
    return m, qo.range_search_products(RP[f"{tag}|w"], m["in_dtype"], spec.steps[0], ratios, m["degree"], xs,
                                       bias=RP[f"{tag}|bias"], objective=m["resolved_objective"],
                                       granularity=m["resolved_granularity"],
                                       element_batch_size=m["element_batch_size"])


def clear_winner(m, tag, want_err) -> np.ndarray:
    """Per searched group: do the reference's two best candidates differ by more than one product landing on the
    neighbouring value of the 16-bit type can change a group's error?  With u the spacing of the type at the group's
    largest product and d = (second-best error)^(1/degree) a bound on any single difference, that change is at most
    (d + u)^degree - d^degree."""
    w = RP[f"{tag}|w"].astype(np.float64)
    bits = {"bf16": 7, "f16": 10}[m["in_dtype"]]
    errs = want_err.reshape(want_err.shape[0], -1)
    groups = errs.shape[1]
    xs = np.concatenate([RP[f"{tag}|x{i}"].reshape(-1, w.shape[1]) for i in range(2)]).astype(np.float64)
    if m["resolved_objective"] == "ProductsError" and groups > 1:
        cg = groups // w.shape[0] if m["resolved_granularity"] == "Group" else groups
        g = w.shape[1] // cg
        y = np.einsum("ncg,rcg->rcn", xs.reshape(-1, cg, g), w.reshape(w.shape[0], cg, g))
        top = np.abs(y).max(axis=2) if m["resolved_granularity"] == "Group" else np.abs(y).max(axis=(0, 2))
        top = top.reshape(-1)
    else:
        top = np.full(groups, np.abs(xs @ w.T).max())
    spacing = 2.0 ** (np.floor(np.log2(np.maximum(top, 1e-30))) - bits)
    srt = np.sort(errs, axis=0)
    d = srt[1] ** (1.0 / m["degree"])
    return (srt[1] - srt[0]) > ((d + spacing) ** m["degree"] - d ** m["degree"])


def check_against_fixture(m, tag, errors, ratio, vmax):
    want_err = RP[f"{tag}|s0|errors"].reshape(np.asarray(errors).shape)
    want_ratio = RP[f"{tag}|s0|ratio"].reshape(ratio.shape)
    errors = np.asarray(errors)
    if m["in_dtype"] == "f32":
        np.testing.assert_allclose(errors, want_err, rtol=2e-3, atol=1e-12)
        srt = np.sort(want_err.reshape(want_err.shape[0], -1), axis=0)
        clear = (srt[1] - srt[0]) > 1e-2 * srt[0]
        same = ratio.reshape(-1) == want_ratio.reshape(-1)
        assert same[clear].all()
        assert same.mean() > 0.97
        if same.all():
            np.testing.assert_array_equal(vmax, RP[f"{tag}|s0|max"])
        return
    # 16-bit products.  The reference's GEMM rounds each product to the 16-bit type after a float32 sum in an order of
    # its backend's choosing; here every product is the correctly rounded sum.  The two can differ by one unit in the
    # last place of a product, so: most entries agree closely, the totals per candidate agree, and the winner is the
    # reference's wherever its best two candidates are further apart than ONE product moving by one unit can explain.
    rel = np.abs(errors - want_err) / np.maximum(want_err, 1e-30)
    assert np.median(rel) < 2e-2
    np.testing.assert_allclose(errors.reshape(errors.shape[0], -1).sum(1), want_err.reshape(errors.shape[0], -1).sum(1),
                               rtol=3e-2)
    if ratio.size > 1:
        clear = clear_winner(m, tag, want_err)
        same = ratio.reshape(-1) == want_ratio.reshape(-1)
        assert same[clear].all(), (int(clear.sum()), int((~same[clear]).sum()))
        assert np.abs(ratio.reshape(-1) - want_ratio.reshape(-1)).mean() < 0.05
        return float(clear.mean())


@pytest.mark.parametrize("tag", sorted(t for t, m in RP.meta.items() if m["reference_raised"] is None))
def test_products_search_matches_reference(tag):
    m, got = run_oracle(tag)
    check_against_fixture(m, tag, got["errors"], got["ratio"], got["max"])


def test_outputs_error_on_a_linear_becomes_products_error():
    tags = [t for t, m in RP.meta.items() if m["objective"] == "OutputsError" and m["granularity"] == "Group"]
    assert tags and all(RP.meta[t]["resolved_objective"] == "ProductsError" for t in tags)


def test_channel_group_products_first_candidates():
    tag = next(t for t, m in RP.meta.items() if m["granularity"] == "ChannelGroup")
    assert "invalid for input of size" in RP.meta[tag]["reference_raised"]
    m, got = run_oracle(tag)
    want = RP[f"{tag}|s0|errors"]
    np.testing.assert_allclose(np.stack(got["errors"][: want.shape[0]]), want, rtol=2e-3)
