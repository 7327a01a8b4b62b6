"""The linear patches (ConcatLinear, ShiftedLinear) against the reference's modules; plain torch modules, so these
checks run without a GPU."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from conftest import load_golden

from deepcompressor_amd.nn.patch import ConcatLinear, ScaleDotProductAttention, ShiftedLinear

PL = load_golden("patch_linear")
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def linear_of(dt):
    lin = nn.Linear(96, 40, bias=True).to(DT[dt])
    lin.weight.data = torch.from_numpy(PL[f"{dt}|w"]).to(DT[dt])
    lin.bias.data = torch.from_numpy(PL[f"{dt}|b"]).to(DT[dt])
    return lin, torch.from_numpy(PL[f"{dt}|x"]).to(DT[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_concat_linear_matches_reference(dt):
    lin, x = linear_of(dt)
    splits = [32, 16]
    cat = ConcatLinear.from_linear(lin, splits)
    m = PL.meta[f"{dt}|concat"]
    assert cat.in_features_list == m["widths"] == splits  # the caller's list is extended, as in the reference
    assert [p.bias is not None for p in cat.linears] == m["bias_on"]
    for i, part in enumerate(cat.linears):
        np.testing.assert_array_equal(part.weight.data.float().numpy(), PL[f"{dt}|concat|w{i}"])
    with torch.no_grad():
        np.testing.assert_array_equal(cat(x).float().numpy(), PL[f"{dt}|concat|y"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_shifted_linear_matches_reference_and_keeps_the_function(dt):
    lin, x = linear_of(dt)
    sh = ShiftedLinear.from_linear(lin, 0.17)
    np.testing.assert_array_equal(sh.shift.float().numpy().reshape(-1), PL[f"{dt}|shift|shift"].reshape(-1))
    np.testing.assert_array_equal(sh.linear.bias.data.float().numpy(), PL[f"{dt}|shift|bias"])
    with torch.no_grad():
        np.testing.assert_array_equal(sh(x).float().numpy(), PL[f"{dt}|shift|y"])
        if dt == "f32":
            assert torch.allclose(sh(x), lin(x), atol=1e-5)


def test_shifted_linear_per_channel_block_shift():
    lin, x = linear_of("f32")
    shift = torch.tensor([0.1, -0.2, 0.3])  # one value per 32 consecutive channels
    sh = ShiftedLinear.from_linear(lin, shift)
    assert sh.shift.shape == (96,) and sh.linear.shifted
    with torch.no_grad():
        assert torch.allclose(sh(x), lin(x), atol=1e-5)


def test_attention_product_module():
    q, k, v = (torch.randn(2, 3, 5, 8) for _ in range(3))
    got = ScaleDotProductAttention()(q, k, v, is_causal=True)
    assert torch.equal(got, torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True))
