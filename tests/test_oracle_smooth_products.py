"""The CPU oracle's products-scored smoothing search (one winner per group of input channels) against fixtures
captured from the reference's SmoothLinearCalibrator."""

import numpy as np
import pytest
from conftest import load_golden

import quant_oracle as qo
import smooth_oracle as so

SP = load_golden("smooth_products")
W_SPEC = dict(dtype="sint4", zero=None, groups=((1, 16),), sdt=("f16",))
X_SPEC = dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",))


def case_inputs(tag):
    m = SP.meta[tag]
    weights = [SP[f"{tag}|w{i}"] for i in range(len(m["rows"]))]
    xs = [SP[f"{tag}|x{i}"] for i in range(3)]
    return m, weights, xs


@pytest.mark.parametrize("tag", sorted(SP.meta))
def test_candidate_errors_match_reference(tag):
    m, weights, xs = case_inputs(tag)
    want = SP[f"{tag}|errors"]  # [candidates, weights, groups]
    assert want.shape == (m["num_candidates"], len(weights), 4)
    x_spec = qo.spec_from_case(X_SPEC) if m["quantize_inputs"] else None
    tol = 2e-3 if m["dtype"] == "f32" else 2e-2
    for c in range(0, m["num_candidates"], 3):
        got = so.products_errors(weights, xs, SP[f"{tag}|scales"][c], qo.spec_from_case(W_SPEC), x_spec, m["dtype"], 16)
        for j in range(len(weights)):
            assert np.all(np.abs(got[j] - want[c, j]) <= tol * want[c, j]), (c, j, got[j] / want[c, j])


@pytest.mark.parametrize("tag", sorted(SP.meta))
def test_best_scale_follows_from_reference_errors(tag):
    """The selection rule alone, exact: replaying the reference's own error tensors must give its final scale."""
    m, weights, _ = case_inputs(tag)
    errors, scales = SP[f"{tag}|errors"], SP[f"{tag}|scales"]
    best_e = best_s = None
    for c in range(m["num_candidates"]):
        best_e, best_s = so.update_best(best_e, best_s, [errors[c, j].copy() for j in range(len(weights))], scales[c],
                                        64, m["num_heads"], m["num_head_repeats"])
    np.testing.assert_array_equal(best_s, SP[f"{tag}|best_scale"])


def test_shared_heads_switch_together():
    m, _, _ = case_inputs("products_chgroup_gqa")
    best = SP["products_chgroup_gqa|best_scale"].reshape(2, 2, 16)
    scales = SP["products_chgroup_gqa|scales"].reshape(m["num_candidates"], 2, 2, 16)
    for u in range(2):  # both query heads of a key/value head take their scale from the same candidate
        picks = [np.flatnonzero([np.array_equal(scales[c, u, r], best[u, r]) for c in range(m["num_candidates"])])
                 for r in range(2)]
        assert set(picks[0]) & set(picks[1])
