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


# ---- convolution counterparts ---------------------------------------------------------------------------------------
from deepcompressor_amd.nn.patch import ConcatConv2d, ShiftedConv2d  # noqa: E402


def conv_of(dt, **kw):
    conv = nn.Conv2d(12, 10, 3, **kw).to(DT[dt])
    conv.weight.data = torch.from_numpy(PL[f"{dt}|conv|w"]).to(DT[dt])
    if conv.bias is not None:
        conv.bias.data = torch.from_numpy(PL[f"{dt}|conv|b"]).to(DT[dt])
    return conv, torch.from_numpy(PL[f"{dt}|conv|x"]).to(DT[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_concat_conv_matches_reference(dt):
    conv, x = conv_of(dt, stride=1, padding=1, bias=True)
    cat = ConcatConv2d.from_conv2d(conv, [4, 5])
    m = PL.meta[f"{dt}|conv|concat"]
    assert cat.in_channels_list == m["widths"] and [p.bias is not None for p in cat.convs] == m["bias_on"]
    with torch.no_grad():
        np.testing.assert_array_equal(cat(x).float().numpy(), PL[f"{dt}|conv|concat|y"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_shifted_conv_matches_reference_and_keeps_the_function(dt):
    conv, x = conv_of(dt, stride=1, padding=1, bias=True)
    sh = ShiftedConv2d.from_conv2d(conv, 0.21)
    assert sh.conv.shifted and sh.padding_mode == "constant" and sh.conv.padding == "valid"
    np.testing.assert_array_equal(sh.conv.bias.data.float().numpy(), PL[f"{dt}|conv|shift|bias"])
    with torch.no_grad():
        np.testing.assert_array_equal(sh(x).float().numpy(), PL[f"{dt}|conv|shift|y"])
        if dt == "f32":
            assert torch.allclose(sh(x), conv(x), atol=1e-5)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_shifted_conv_block_shift(dt):
    bare, x = conv_of(dt, stride=2, padding=0, bias=False)
    shift = torch.tensor([0.1, -0.2, 0.3])
    direct = ShiftedConv2d(12, 10, 3, shift=shift, stride=2, padding=0, bias=True, dtype=DT[dt])
    direct.conv.weight.data.copy_(bare.weight.data)
    direct.conv.bias.data.zero_()
    np.testing.assert_array_equal(direct.shift.float().numpy(), PL[f"{dt}|conv|blockshift|shift"])
    with torch.no_grad():
        np.testing.assert_array_equal(direct(x).float().numpy(), PL[f"{dt}|conv|blockshift|y"])
        if dt == "f32":  # from_conv2d takes the block shift here (the reference's trips over the tuple padding)
            assert torch.allclose(ShiftedConv2d.from_conv2d(bare, shift)(x), bare(x), atol=1e-5)
