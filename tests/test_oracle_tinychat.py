"""The CPU oracle's TinyChat converter against fixtures captured from the reference's converter."""

import numpy as np
import pytest
from conftest import load_golden

import tinychat_oracle as tc

TC = load_golden("tinychat")
CASES = sorted(t for t in TC.meta if "|" not in t and t != "ceil_num_groups")
ADANORM = sorted(t.split("|")[1] for t in TC.meta if t.startswith("w4x16|"))


def inputs_of(tag, module):
    p = f"{module}.weight"
    pre = f"{tag}|scale|{p}.zero" in TC
    zero = TC[f"{tag}|scale|{p}.zero" if pre else f"{tag}|scale|{p}.scaled_zero"]
    return TC[f"{tag}|model|{p}"], TC[f"{tag}|scale|{p}.scale.0"], zero, pre


def modules_of(tag):
    return sorted({k.split("|")[2][: -len(".qweight")] for k in TC.arrays.files
                   if k.startswith(f"{tag}|out|") and k.endswith(".qweight")})


def test_ceil_num_groups():
    for key, want in TC.meta["ceil_num_groups"].items():
        ic, gs = map(int, key.split("|"))
        assert tc.ceil_num_groups(ic, gs) == want, key


def test_pack_w4():
    np.testing.assert_array_equal(tc.pack_w4(TC["pack|codes"]), TC["pack|packed"])


@pytest.mark.parametrize("tag", CASES)
def test_converter_matches_reference(tag):
    for module in modules_of(tag):
        weight, scale, zero, pre = inputs_of(tag, module)
        w, s, z = tc.convert(weight, scale, zero, TC.meta[tag]["dtype"], zero_pre_scaled=pre)
        np.testing.assert_array_equal(w, TC[f"{tag}|out|{module}.qweight"])
        np.testing.assert_array_equal(s, TC[f"{tag}|out|{module}.scales"])
        np.testing.assert_array_equal(z, TC[f"{tag}|out|{module}.scaled_zeros"])


@pytest.mark.parametrize("tag", ADANORM)
def test_nunchaku_w4x16_matches_reference(tag):
    m = TC.meta[f"w4x16|{tag}"]
    w, s, z, b = tc.convert_w4x16(TC[f"w4x16|{tag}|weight"], TC[f"w4x16|{tag}|scale"], TC[f"w4x16|{tag}|bias"], m["splits"])
    assert list(w.shape) == m["out_weight"][1]
    np.testing.assert_array_equal(w, TC[f"w4x16|{tag}|out_weight"])
    np.testing.assert_array_equal(s, TC[f"w4x16|{tag}|out_scale"])
    np.testing.assert_array_equal(z, TC[f"w4x16|{tag}|out_zero"])
    np.testing.assert_array_equal(b, TC[f"w4x16|{tag}|out_bias"])
