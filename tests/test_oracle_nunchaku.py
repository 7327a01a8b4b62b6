"""The CPU oracle's Nunchaku converter against fixtures captured from the reference's converter."""

import numpy as np
import pytest
from conftest import load_golden

import nunchaku_oracle as nk

NK = load_golden("nunchaku")
LINEARS = sorted(t for t in NK.meta if not t.startswith("w8"))


def linear_inputs(tag):
    pre = f"{tag}|in|"
    kw = {k[len(pre):]: NK[k] for k in NK.arrays.files if k.startswith(pre)}
    if "lora_down" in kw:
        kw["lora"] = (kw.pop("lora_down"), kw.pop("lora_up"))
    m = NK.meta[tag]
    kw["float_point"], kw["smooth_fused"] = m["float_point"], m["smooth_fused"]
    return kw


@pytest.mark.parametrize("bits", [4, 8])
def test_pack_weight(bits):
    np.testing.assert_array_equal(nk.pack_weight(NK[f"pack{bits}|codes"], bits), NK[f"pack{bits}|packed"])


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_scale_and_lowrank_packs(dt):
    s = NK[f"scale|{dt}|in"]
    np.testing.assert_array_equal(nk.pack_scale(s, 64), NK[f"scale|{dt}|g64"])
    np.testing.assert_array_equal(nk.pack_scale(s[:, :, :1], -1), NK[f"scale|{dt}|chn"])
    np.testing.assert_array_equal(nk.pack_micro_scale(NK[f"micro|{dt}|in"], 16), NK[f"micro|{dt}|g16"])
    for side, down in (("down", True), ("up", False)):
        w = NK[f"lowrank|{dt}|{side}"]
        packed = nk.pack_lowrank(nk.pad(w, (128,), (1 if down else 0,)), down)
        np.testing.assert_array_equal(packed, NK[f"lowrank|{dt}|{side}_packed"])
        np.testing.assert_array_equal(nk.unpack_lowrank(packed, down), NK[f"lowrank|{dt}|{side}_unpacked"])


def test_e4m3_encoder_is_exact_on_representable_values_and_rounds_to_even():
    table = nk._e4m3_table()
    np.testing.assert_array_equal(nk.to_e4m3fn(table.astype(np.float32)), np.arange(127, dtype=np.uint8))
    np.testing.assert_array_equal(nk.to_e4m3fn(-table[1:].astype(np.float32)), np.arange(1, 127, dtype=np.uint8) | 0x80)
    mid = ((table[:-1] + table[1:]) / 2).astype(np.float32)  # exact in float32
    np.testing.assert_array_equal(nk.to_e4m3fn(mid) % 2, 0)


@pytest.mark.parametrize("tag", LINEARS)
def test_linear_conversion_matches_reference(tag):
    m = NK.meta[tag]
    out = nk.linear_state_dict(dt=m["dtype"], **linear_inputs(tag))
    assert sorted(out) == sorted(m["out"])
    for k, (_, shape) in m["out"].items():
        assert list(out[k].shape) == shape, k
        np.testing.assert_array_equal(out[k], NK[f"{tag}|out|{k}"], err_msg=k)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_w8_conversion_matches_reference(dt):
    w, s, b = nk.convert_w8(NK[f"w8|{dt}|in|weight"], NK[f"w8|{dt}|in|scale"], NK[f"w8|{dt}|in|bias"])
    np.testing.assert_array_equal(w, NK[f"w8|{dt}|out|weight"])
    np.testing.assert_array_equal(s, NK[f"w8|{dt}|out|scale"])
    np.testing.assert_array_equal(b, NK[f"w8|{dt}|out|bias"])
