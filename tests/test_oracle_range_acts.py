"""The CPU oracle's activation range search (inputs / outputs of a module, static and dynamic) against fixtures
captured from the reference."""

import numpy as np
import pytest
from _toy import toy_gated_numpy
from conftest import load_golden

import quant_oracle as qo

class _Merged:
    """The activation range fixtures: tensor / outputs scored searches and the products scored ones."""

    def __init__(self, *parts):
        self.parts = parts
        self.meta = dict(cases=parts[0].meta["cases"], entries={k: v for p in parts for k, v in p.meta["entries"].items()})

    def _part(self, key):
        return next(p for p in self.parts if key in p)

    def __contains__(self, key):
        return any(key in p for p in self.parts)

    def __getitem__(self, key):
        return self._part(key)[key]


RA = _Merged(load_golden("range_acts"), load_golden("range_acts_products"))
ENTRIES = RA.meta["entries"]
ULP = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11, "f32": 2.0 ** -22}


def run_oracle(tag):
    m = ENTRIES[tag]
    spec = qo.spec_from_case(RA.meta["cases"][m["case"]])
    ratios = list(RA[f"{tag}|ratios"])
    xs = [RA[f"{tag}|x{i}"] for i in range(3)]
    weights = {k: RA[f"{tag}|w_{k}"] for k in ("up", "gate", "down")}
    mid_of, evaluate = toy_gated_numpy(weights, m["in_dtype"])
    inputs = m["tensor_type"] == "Inputs"
    objective = m["objective"] if inputs else "OutputsError"
    granularity = m["granularity"] if inputs and objective != "OutputsError" else "Layer"
    tensors = xs if inputs else [mid_of(x) for x in xs]
    batches = [t.reshape(-1, t.shape[-1]) for t in tensors]
    ev = (lambda fn: evaluate(xs, fn_in=fn)) if inputs else (lambda fn: evaluate(xs, fn_mid=fn))
    return m, qo.range_search_acts(batches, m["in_dtype"], spec.steps[0], ratios, m["degree"], objective=objective,
                                   granularity=granularity, static=m["static"], allow_scale=m["allow_scale"],
                                   evaluate=ev, weights=[weights["up"], weights["gate"]])


def check(m, tag, got, *, rtol_err):
    want = RA[f"{tag}|s0|errors"]
    errs = np.stack(got["errors"]).reshape(want.shape)
    np.testing.assert_allclose(errs, want, rtol=rtol_err)
    # the same winner wherever the reference's best two candidates are further apart than the tolerance
    srt = np.sort(want.reshape(want.shape[0], -1), axis=0)
    clear = (srt[1] - srt[0]) > 4 * rtol_err * srt[0] if want.shape[0] > 1 else np.ones(srt.shape[1], dtype=bool)
    want_ratio = RA[f"{tag}|s0|ratio"].reshape(-1)
    same = got["ratio"].reshape(-1) == want_ratio
    assert same[clear].all(), (got["ratio"], want_ratio)
    if same.all() and f"{tag}|s0|max" in RA:
        # inputs: the cached tensor is the fixture itself, so the range is exact; outputs: the tensor is recomputed
        # here with float64 matrix products and can sit one unit in the last place away from the reference's
        rtol = 0 if m["tensor_type"] == "Inputs" else 2 * ULP[m["in_dtype"]]
        np.testing.assert_allclose(got["max"], RA[f"{tag}|s0|max"], rtol=rtol, atol=0)
        if f"{tag}|s0|min" in RA:
            np.testing.assert_allclose(got["min"], RA[f"{tag}|s0|min"], rtol=rtol, atol=0)
    if f"{tag}|s0|out_ratio" in RA:  # ratio search: the stored result is the ratio alone
        assert got["max"] is None and f"{tag}|s0|max" not in RA


def tolerance(m) -> float:
    """TensorError: one unit in the last place of the tensor's dtype per batch sum (multi-axis sums have no pinned
    order).  OutputsError: float32 modules within 1e-3; 16-bit modules statistically only (an intermediate landing on
    the neighbouring 16-bit value is as large as the quantization effect being measured)."""
    if m["resolved_objective"] == "TensorError":
        return 4 * ULP[m["in_dtype"]]
    if m["resolved_objective"] == "ProductsError":  # float32 sums of many products in an order that is not pinned
        return 1e-3 if m["in_dtype"] == "f32" else 2e-2
    return 1e-3 if m["in_dtype"] == "f32" else 0.15


@pytest.mark.parametrize("tag", sorted(t for t, m in ENTRIES.items() if m["reference_raised"] is None))
def test_activation_search_matches_reference(tag):
    m, got = run_oracle(tag)
    check(m, tag, got, rtol_err=tolerance(m))


@pytest.mark.parametrize("tag", sorted(t for t, m in ENTRIES.items() if m["reference_raised"] is not None))
def test_group_granularity_on_inputs_raises_like_the_reference(tag):
    assert "Unknown granularity" in ENTRIES[tag]["reference_raised"]
    with pytest.raises(ValueError, match="Unknown granularity"):
        run_oracle(tag)
