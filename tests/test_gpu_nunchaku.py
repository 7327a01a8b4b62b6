"""Nunchaku conversion on the GPU (pack kernel + converters) against the reference's converter outputs, and a
full-size pack against the oracle."""

import numpy as np
import pytest
import torch
from test_oracle_nunchaku import LINEARS, NK, linear_inputs

import nunchaku_oracle as nk
from deepcompressor_amd.backend.nunchaku import (NunchakuWeightPacker, convert_to_nunchaku_w4x4y16_linear_state_dict,
                                                 convert_to_nunchaku_w8x8y16_linear_weight)

pytestmark = pytest.mark.gpu
DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def host(t: torch.Tensor) -> np.ndarray:
    t = t.cpu()
    if t.dtype == torch.float8_e4m3fn:
        return t.view(torch.uint8).numpy()
    return t.numpy() if t.dtype == torch.int8 else t.float().numpy()


@pytest.mark.parametrize("bits", [4, 8])
def test_pack_kernel_matches_reference(bits):
    got = NunchakuWeightPacker(bits=bits).pack_weight(torch.from_numpy(NK[f"pack{bits}|codes"]).cuda())
    np.testing.assert_array_equal(host(got), NK[f"pack{bits}|packed"])


@pytest.mark.parametrize("bits,shape", [(4, (3072, 12288)), (8, (3072, 3072))])
def test_pack_kernel_flux_size_matches_oracle(bits, shape):
    g = torch.Generator().manual_seed(bits)
    half = 1 << (bits - 1)
    codes = torch.randint(-half, half, shape, generator=g, dtype=torch.int32)
    got = NunchakuWeightPacker(bits=bits).pack_weight(codes.cuda())
    np.testing.assert_array_equal(host(got), nk.pack_weight(codes.numpy(), bits))


def test_pack_rejects_ragged_shapes_and_wrong_dtype():
    with pytest.raises(AssertionError, match="divisible"):
        NunchakuWeightPacker(bits=4).pack_weight(torch.zeros(64, 128, dtype=torch.int32, device="cuda"))
    with pytest.raises(AssertionError, match="divisible"):
        NunchakuWeightPacker(bits=4).pack_weight(torch.zeros(128, 64, dtype=torch.int32, device="cuda"))
    with pytest.raises(AssertionError, match="int32"):
        NunchakuWeightPacker(bits=4).pack_weight(torch.zeros(128, 128, dtype=torch.int8, device="cuda"))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_scale_and_lowrank_packs(dt):
    packer = NunchakuWeightPacker(bits=4)
    t = lambda key: torch.from_numpy(NK[key]).to(DT[dt]).cuda()  # noqa: E731
    s = t(f"scale|{dt}|in")
    np.testing.assert_array_equal(host(packer.pack_scale(s, 64)), NK[f"scale|{dt}|g64"])
    np.testing.assert_array_equal(host(packer.pack_scale(s[:, :, :1].contiguous(), -1)), NK[f"scale|{dt}|chn"])
    np.testing.assert_array_equal(host(packer.pack_scale(t(f"micro|{dt}|in"), 16)), NK[f"micro|{dt}|g16"])
    for side, down in (("down", True), ("up", False)):
        packed = packer.pack_lowrank_weight(packer.pad_lowrank_weight(t(f"lowrank|{dt}|{side}"), down), down)
        np.testing.assert_array_equal(host(packed), NK[f"lowrank|{dt}|{side}_packed"])
        np.testing.assert_array_equal(host(packer.unpack_lowrank_weight(packed, down)),
                                      NK[f"lowrank|{dt}|{side}_unpacked"])


@pytest.mark.parametrize("tag", LINEARS)
def test_linear_conversion_matches_reference(tag):
    m = NK.meta[tag]
    kw = linear_inputs(tag)
    to = lambda a: torch.from_numpy(a).to(DT[m["dtype"]]).cuda()  # noqa: E731
    for k, v in list(kw.items()):
        if isinstance(v, np.ndarray):
            kw[k] = to(v)
        elif isinstance(v, tuple):
            kw[k] = tuple(to(a) for a in v)
    out = convert_to_nunchaku_w4x4y16_linear_state_dict(**kw)
    assert sorted(out) == sorted(m["out"])
    for k, (dtype, shape) in m["out"].items():
        assert str(out[k].dtype) == dtype and list(out[k].shape) == shape and out[k].is_cuda, k
        np.testing.assert_array_equal(host(out[k]), NK[f"{tag}|out|{k}"], err_msg=k)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_w8_conversion_matches_reference(dt):
    t = lambda key: torch.from_numpy(NK[f"w8|{dt}|in|{key}"]).to(DT[dt]).cuda()  # noqa: E731
    w, s, b = convert_to_nunchaku_w8x8y16_linear_weight(t("weight"), t("scale"), t("bias"))
    for name, got in (("weight", w), ("scale", s), ("bias", b)):
        np.testing.assert_array_equal(host(got), NK[f"w8|{dt}|out|{name}"], err_msg=name)


@pytest.mark.parametrize("tag", LINEARS[:2])
def test_pointwise_convolution_weight_converts_like_its_matrix(tag):
    """A [out][in][1][1] weight is flattened first, and its scale is checked against the flattened weight's rank as
    the reference does (backend/nunchaku/convert.py:24-30, 37); the quantizer's own scale of such a weight (one more
    (1, 1) pair per kernel dimension) is taken as well.  Same packed tensors as the linear either way."""
    m = NK.meta[tag]
    kw = linear_inputs(tag)
    to = lambda a: torch.from_numpy(a).to(DT[m["dtype"]]).cuda()  # noqa: E731
    for k, v in list(kw.items()):
        if isinstance(v, np.ndarray):
            kw[k] = to(v)
        elif isinstance(v, tuple):
            kw[k] = tuple(to(a) for a in v)
    want = convert_to_nunchaku_w4x4y16_linear_state_dict(**kw)
    conv = dict(kw, weight=kw["weight"][:, :, None, None])
    got = convert_to_nunchaku_w4x4y16_linear_state_dict(**conv)  # 4-D scale, as the reference accepts
    assert sorted(got) == sorted(want) and all(torch.equal(got[k], want[k]) for k in want)
    if kw["scale"].numel() > 1:
        conv["scale"] = kw["scale"].reshape(*kw["scale"].shape, 1, 1, 1, 1)  # 8-D: the quantizer's view of the 4-D weight
        got = convert_to_nunchaku_w4x4y16_linear_state_dict(**conv)
        assert all(torch.equal(got[k], want[k]) for k in want)
