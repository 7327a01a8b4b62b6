"""The CPU oracle's power-of-two (exponent-only) scale levels against reference fixtures."""

import numpy as np
import pytest
from conftest import assert_bit_equal, load_golden

import quant_oracle as qo

ES = load_golden("exp_scale")


@pytest.mark.parametrize("tag", sorted(ES.meta["entries"]))
def test_exponent_scale_levels_match_reference(tag):
    m = ES.meta["entries"][tag]
    spec = qo.spec_from_case(ES.meta["cases"][m["case"]])
    out = qo.fake_quantize(ES[f"{tag}|x"], m["in_dtype"], spec)
    assert_bit_equal(out["scale"], ES[f"{tag}|scale"].reshape(out["scale"].shape), "compound scale")
    leaves = sorted(k for k in ES.arrays.files if k.startswith(f"{tag}|leaf|"))
    assert len(leaves) == len(out["scale_leaves"])
    for key, leaf in zip(leaves, out["scale_leaves"]):
        assert_bit_equal(leaf, ES[key].reshape(leaf.shape), key)
        if ES.meta["cases"][m["case"]]["sdt"][int(key.rsplit(".", 1)[1])].startswith("uexp"):
            assert np.array_equal(np.log2(leaf), np.rint(np.log2(leaf))), "a power-of-two level"
    assert_bit_equal(np.broadcast_to(out["zero"], ES[f"{tag}|zero"].shape), ES[f"{tag}|zero"], "zero")
    assert_bit_equal(out["codes"], ES[f"{tag}|codes"], "codes")
    assert_bit_equal(out["dequant"], ES[f"{tag}|dequant"], "dequant")
