"""The CPU oracle's smoothing search scored on the weights alone (TensorError) against fixtures captured from the
reference's SmoothLinearCalibrator."""

import numpy as np
import pytest
from conftest import load_golden

import quant_oracle as qo
import smooth_oracle as so

ST = load_golden("smooth_tensor")
W_SPEC = dict(dtype="sint4", zero=None, groups=((1, 16),), sdt=("f16",))
# 16-bit errors are float32 sums rounded to the weights' dtype once: a total on a rounding edge may land one unit off;
# float32 errors are sums of 640+ float32 terms in an order that is not pinned
ULPS = {"f32": 2e-6, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}


def case_inputs(tag):
    m = ST.meta[tag]
    return m, [ST[f"{tag}|w{i}"] for i in range(len(m["rows"]))], [ST[f"{tag}|x{i}"] for i in range(3)]


@pytest.mark.parametrize("tag", sorted(ST.meta))
def test_candidate_errors_match_reference(tag):
    m, weights, _ = case_inputs(tag)
    want = ST[f"{tag}|errors"]  # [candidates, weights, groups or 1]
    for c in range(m["num_candidates"]):
        got = so.tensor_errors(weights, ST[f"{tag}|scales"][c], qo.spec_from_case(W_SPEC), m["dtype"], 16,
                               granularity=m["granularity"])
        for j in range(len(weights)):
            assert np.all(np.abs(got[j] - want[c, j]) <= ULPS[m["dtype"]] * want[c, j]), (c, j, got[j], want[c, j])


@pytest.mark.parametrize("tag", sorted(ST.meta))
def test_best_scale_follows_from_reference_errors(tag):
    m, weights, _ = case_inputs(tag)
    errors, scales = ST[f"{tag}|errors"], ST[f"{tag}|scales"]
    best_e = best_s = None
    for c in range(m["num_candidates"]):
        best_e, best_s = so.update_best(best_e, best_s, [errors[c, j].copy() for j in range(len(weights))], scales[c],
                                        64, m["num_heads"], m["num_head_repeats"])
    np.testing.assert_array_equal(best_s.reshape(ST[f"{tag}|best_scale"].shape), ST[f"{tag}|best_scale"])
