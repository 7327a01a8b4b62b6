"""The CPU oracle's rounding-offset path (codes = clamp(floor(t) + round_delta)) against reference fixtures."""

import pytest
from conftest import assert_bit_equal, load_golden

import quant_oracle as qo

FQ = load_golden("fake_quant")
RD = load_golden("round_delta")


@pytest.mark.parametrize("tag", sorted(RD.meta))
def test_round_delta_matches_reference(tag):
    m = RD.meta[tag]
    spec = qo.spec_from_case(FQ.meta["cases"][m["case"]])
    out = qo.fake_quantize(RD[f"{tag}|x"], m["in_dtype"], spec, round_delta=RD[f"{tag}|delta"])
    assert_bit_equal(out["codes"], RD[f"{tag}|codes"], "codes")
    assert_bit_equal(out["dequant"], RD[f"{tag}|dequant"], "dequant")
