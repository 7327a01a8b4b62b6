"""QServe conversion on the GPU (pack kernel + converter) against the reference's converter outputs, and the whole
chain weight stage -> scale.pt / model.pt -> QServe tensors against the oracle."""

import numpy as np
import pytest
import torch
from test_oracle_qserve import QS, inputs_of, modules_of

import qserve_oracle as qs
from deepcompressor_amd.backend.qserve import (QServePacker, convert_to_qserve_state_dict,
                                               convert_to_qserve_w4x8y16_linear_weight)

pytestmark = pytest.mark.gpu


def test_pack_kernel_matches_reference():
    got = QServePacker().pack_weight(torch.from_numpy(QS["pack|codes"]).cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), QS["pack|packed"])


def test_pack_kernel_large_matches_oracle():
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, 16, (4096, 4096), generator=g, dtype=torch.uint8)
    got = QServePacker().pack_weight(codes.cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, qs.pack_weight(codes.numpy()))


def test_pack_rejects_ragged_shapes():
    with pytest.raises(AssertionError, match="divisible"):
        QServePacker().pack_weight(torch.zeros(48, 64, dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("tag", sorted(QS.meta))
def test_state_dict_conversion_matches_reference(tag):
    m = QS.meta[tag]
    state = {k.split("|model|")[1]: torch.from_numpy(QS[k]).to(torch.float16) for k in QS.arrays.files
             if k.startswith(f"{tag}|model|")}
    scales = {k.split("|scale|")[1]: torch.from_numpy(QS[k]) for k in QS.arrays.files if k.startswith(f"{tag}|scale|")}
    out = convert_to_qserve_state_dict(state, scales, m["bits"])
    assert sorted(out) == sorted(m["out"])
    for k, (dtype, shape) in m["out"].items():
        got = out[k]
        assert str(got.dtype) == dtype and list(got.shape) == shape and got.device.type == "cpu", k
        want = QS[f"{tag}|out|{k}"]
        np.testing.assert_array_equal(got.numpy() if got.dtype == torch.int8 else got.float().numpy(), want, err_msg=k)


def test_single_weight_conversion_on_gpu():
    tag = "w4_g128_post"
    for module in modules_of(tag):
        weight, scale, zero, subscale, pre = inputs_of(tag, module)
        t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
        w, s, z, ss = convert_to_qserve_w4x8y16_linear_weight(t(weight).to(torch.float16), t(scale), t(zero), t(subscale),
                                                              zero_pre_scaled=pre)
        assert w.is_cuda and w.dtype == torch.int8
        np.testing.assert_array_equal(w.cpu().numpy(), QS[f"{tag}|out|{module}.qweight"])
        np.testing.assert_array_equal(ss.cpu().numpy(), QS[f"{tag}|out|{module}.s2_scales"])
        np.testing.assert_array_equal(z.cpu().numpy(), QS[f"{tag}|out|{module}.s2_zeros"])
