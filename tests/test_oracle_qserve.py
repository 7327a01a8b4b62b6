"""The CPU oracle's QServe converter against fixtures captured from the reference's converter."""

import numpy as np
import pytest
from conftest import load_golden

import qserve_oracle as qs

QS = load_golden("qserve")


def inputs_of(tag, module):
    p = f"{module}.weight"
    weight = QS[f"{tag}|model|{p}"]
    scale = QS[f"{tag}|scale|{p}.scale.0"]
    sub_key = f"{tag}|scale|{p}.scale.1"
    subscale = QS[sub_key] if sub_key in QS else None
    pre = f"{tag}|scale|{p}.zero" in QS
    zero_key = f"{tag}|scale|{p}.zero" if pre else f"{tag}|scale|{p}.scaled_zero"
    zero = QS[zero_key] if zero_key in QS else None
    return weight, scale, zero, subscale, pre


def modules_of(tag):
    return sorted({k.split("|")[2][: -len(".weight")] for k in QS.arrays.files
                   if k.startswith(f"{tag}|in|")})


def test_pack_weight():
    np.testing.assert_array_equal(qs.pack_weight(QS["pack|codes"]), QS["pack|packed"])


@pytest.mark.parametrize("tag", sorted(QS.meta))
def test_converter_matches_reference(tag):
    m = QS.meta[tag]
    for module in modules_of(tag):
        weight, scale, zero, subscale, pre = inputs_of(tag, module)
        if m["bits"] == 8:
            w, s = qs.convert_w8(weight, scale)
            np.testing.assert_array_equal(w, QS[f"{tag}|out|{module}.weight"])
            np.testing.assert_array_equal(s, QS[f"{tag}|out|{module}.dequant_scale"])
            continue
        w, s, z, ss = qs.convert_w4(weight, scale, zero, subscale, zero_pre_scaled=pre)
        np.testing.assert_array_equal(w, QS[f"{tag}|out|{module}.qweight"])
        np.testing.assert_array_equal(s, QS[f"{tag}|out|{module}.s1_scales"])
        if ss is None:
            np.testing.assert_array_equal(z, QS[f"{tag}|out|{module}.s1_szeros"])
        else:
            np.testing.assert_array_equal(ss, QS[f"{tag}|out|{module}.s2_scales"])
            np.testing.assert_array_equal(z, QS[f"{tag}|out|{module}.s2_zeros"])
