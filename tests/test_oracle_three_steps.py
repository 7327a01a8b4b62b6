"""Three progressive steps (per-row fp16 -> sint8, groups of 256 -> sint6, groups of 64 -> 4 bits): the CPU oracle
against fixtures made by the reference's ``Quantizer`` (``oracle/make_golden.py::gen_three_steps``).  No example
configuration of the reference chains three steps; its ``_quantize`` (quantizer/impl/base.py:211-279) handles any
number, and these fixtures pin that behaviour: only the second-last step takes the protective range, earlier steps are
clamped on the way back.  Bar: bit-exact."""

import numpy as np
import pytest
from conftest import assert_bit_equal, load_golden

import quant_oracle as qo

TS = load_golden("three_steps")


@pytest.mark.parametrize("tag", sorted(TS.meta["entries"]))
def test_three_step_fake_quant_matches_reference(tag):
    e = TS.meta["entries"][tag]
    spec = qo.spec_from_case(TS.meta["cases"][e["case"]])
    assert len(spec.steps) == 3
    dyn = None if e["dyn"] is None else [{"ratio": r} for r in e["dyn"]]
    out = qo.fake_quantize(TS[f"{tag}|x"], e["in_dtype"], spec, dynamic_range=dyn)
    assert list(out["view_shape"]) == e["view_shape"]
    assert_bit_equal(out["scale"], TS[f"{tag}|scale"], "compound scale")
    assert len(out["scale_leaves"]) == e["n_leaves"] == 3
    for i, leaf in enumerate(out["scale_leaves"]):
        assert_bit_equal(leaf, TS[f"{tag}|leaf{i}"], f"scale leaf {i}")
    want_zero = TS[f"{tag}|zero"]
    assert_bit_equal(np.broadcast_to(out["zero"], want_zero.shape), want_zero, "zero")
    assert_bit_equal(out["codes"], TS[f"{tag}|codes"], "codes")
    assert_bit_equal(out["dequant"], TS[f"{tag}|dequant"], "dequant")
