"""TinyChat (AWQ W4A16) conversion on the GPU (pack kernel + converters) against the reference's converter outputs,
and a full-size pack against the oracle."""

import numpy as np
import pytest
import torch
from test_oracle_tinychat import ADANORM, CASES, TC

import tinychat_oracle as tc
from deepcompressor_amd.backend.nunchaku import convert_to_nunchaku_w4x16_linear_weight
from deepcompressor_amd.backend.tinychat import ceil_num_groups, convert_to_tinychat_state_dict, pack_w4

pytestmark = pytest.mark.gpu
DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def host(t: torch.Tensor) -> np.ndarray:
    t = t.cpu()
    return t.numpy() if t.dtype in (torch.int16, torch.int32) else t.float().numpy()


def test_ceil_num_groups():
    for key, want in TC.meta["ceil_num_groups"].items():
        ic, gs = map(int, key.split("|"))
        assert ceil_num_groups(ic, gs) == want, key
    with pytest.raises(NotImplementedError):
        ceil_num_groups(64, 16)


def test_pack_kernel_matches_reference():
    got = pack_w4(torch.from_numpy(TC["pack|codes"]).cuda())
    np.testing.assert_array_equal(host(got), TC["pack|packed"])


def test_pack_kernel_llama_size_matches_oracle():
    g = torch.Generator().manual_seed(5)
    codes = torch.randint(0, 16, (4096, 14336), generator=g, dtype=torch.int32)
    np.testing.assert_array_equal(host(pack_w4(codes.cuda())), tc.pack_w4(codes.numpy()))


def test_pack_rejects_ragged_shapes():
    with pytest.raises(AssertionError, match="divisible by 32"):
        pack_w4(torch.zeros(8, 48, dtype=torch.int32, device="cuda"))
    with pytest.raises(AssertionError, match="4 output x 64 input"):
        pack_w4(torch.zeros(6, 64, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("tag", CASES)
def test_state_dict_conversion_matches_reference(tag):
    m = TC.meta[tag]
    dtype = DT[m["dtype"]]
    state = {k.split("|model|")[1]: torch.from_numpy(TC[k]).to(dtype) for k in TC.arrays.files
             if k.startswith(f"{tag}|model|")}
    scales = {k.split("|scale|")[1]: torch.from_numpy(TC[k]) for k in TC.arrays.files if k.startswith(f"{tag}|scale|")}
    out = convert_to_tinychat_state_dict(state, scales)
    assert sorted(out) == sorted(m["out"])
    for k, (dt, shape) in m["out"].items():
        assert str(out[k].dtype) == dt and list(out[k].shape) == shape and out[k].device.type == "cpu", k
        np.testing.assert_array_equal(host(out[k]), TC[f"{tag}|out|{k}"], err_msg=k)


@pytest.mark.parametrize("tag", ADANORM)
def test_nunchaku_w4x16_matches_reference(tag):
    m = TC.meta[f"w4x16|{tag}"]
    t = lambda key: torch.from_numpy(TC[f"w4x16|{tag}|{key}"]).to(torch.bfloat16).cuda()  # noqa: E731
    w, s, z, b = convert_to_nunchaku_w4x16_linear_weight(t("weight"), t("scale"), bias=t("bias"),
                                                         adanorm_splits=m["splits"])
    assert [str(w.dtype), list(w.shape)] == m["out_weight"]
    for name, got in (("out_weight", w), ("out_scale", s), ("out_zero", z), ("out_bias", b)):
        np.testing.assert_array_equal(host(got), TC[f"w4x16|{tag}|{name}"], err_msg=name)
