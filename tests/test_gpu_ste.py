"""simple_quantize on tensors that require grad: the forward is the HIP rounding kernel, the backward is the
straight-through gradient with the clamps' masks, as autograd gives it for the reference's function."""

import numpy as np
import pytest
import torch
from conftest import assert_bit_equal, load_golden

from deepcompressor_amd.data.dtype import QuantDataType
from deepcompressor_amd.data.range import QuantRange
from deepcompressor_amd.quantizer.impl.simple import simple_quantize
from oracle import quant_oracle as qo

pytestmark = pytest.mark.gpu
ST = load_golden("ste")
TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def run(x, up, dtype_name, zp, qr, delta=None):
    qdt = TORCH_DT[dtype_name] if dtype_name in TORCH_DT else QuantDataType.from_str(dtype_name)
    leaf = torch.from_numpy(x).cuda().requires_grad_(True)
    out = simple_quantize(leaf * 1.0, quant_dtype=qdt, has_zero_point=zp,
                          quant_range=QuantRange(*qr) if qr is not None else None,
                          round_delta=torch.from_numpy(delta).cuda() if delta is not None else None)
    out.backward(torch.from_numpy(up).cuda())
    return out.detach().cpu().numpy(), leaf.grad.cpu().numpy()


@pytest.mark.parametrize("name", sorted(ST.meta))
def test_gradient_matches_reference(name):
    c = ST.meta[name]
    out, grad = run(ST[f"{name}|x"], ST[f"{name}|up"], c["dtype"], c["zp"], tuple(c["qr"]) if "qr" in c else None,
                    ST[f"{name}|delta"] if c.get("delta") else None)
    assert_bit_equal(out, ST[f"{name}|out"], f"{name} forward")
    if c["dtype"].startswith(("uexp", "sexp")):
        np.testing.assert_allclose(grad, ST[f"{name}|grad"], rtol=5e-7, atol=0)  # mask·q/x then ·upstream against torch's ·q·ln2, /(x·ln2)
    else:
        assert_bit_equal(grad, ST[f"{name}|grad"], f"{name} gradient")


def test_float_type_with_range_masks_like_the_oracle():
    # not runnable in the reference (clamp_ on a view made inside its custom Function); checked against the restatement
    g = np.random.default_rng(5)
    x = (g.standard_normal((8, 96)) * 3.5).astype(np.float32)
    up = g.standard_normal((8, 96)).astype(np.float32)
    out, grad = run(x, up, "sfp4_e2m1_all", False, (-3.0, 2.0))
    dt = qo.QDtype.parse("sfp4_e2m1_all")
    assert_bit_equal(out, qo.simple_quantize(x, dt, has_zero_point=False, qr=(-3.0, 2.0)), "forward")
    assert_bit_equal(grad, qo.simple_quantize_grad(x, up, dt, has_zero_point=False, qr=(-3.0, 2.0)), "gradient")


def test_no_grad_path_is_unchanged():
    x = torch.randn(4, 64, device="cuda") * 4
    a = simple_quantize(x, quant_dtype=QuantDataType.from_str("sint4"), has_zero_point=False)
    b = simple_quantize(x.clone().requires_grad_(True) * 1.0, quant_dtype=QuantDataType.from_str("sint4"),
                        has_zero_point=False)
    assert torch.equal(a, b.detach()) and b.requires_grad and not a.requires_grad
