"""Gradient of simple_quantize (straight-through rounding, masking clamps): the numpy restatement against the gradients
autograd produced for the reference's function."""

import numpy as np
import pytest
from conftest import assert_bit_equal, load_golden

from oracle import quant_oracle as qo

ST = load_golden("ste")


def oracle_dtype(name):
    return name if name in ("f16", "bf16", "f32") else qo.QDtype.parse(name)


@pytest.mark.parametrize("name", sorted(ST.meta))
def test_oracle_gradient_matches_reference(name):
    c = ST.meta[name]
    qr = tuple(c["qr"]) if "qr" in c else None
    x, up = ST[f"{name}|x"], ST[f"{name}|up"]
    delta = ST[f"{name}|delta"] if c.get("delta") else None
    out = qo.simple_quantize(x, oracle_dtype(c["dtype"]), has_zero_point=c["zp"], qr=qr, round_delta=delta)
    assert_bit_equal(out, ST[f"{name}|out"], f"{name} forward")
    grad = qo.simple_quantize_grad(x, up, oracle_dtype(c["dtype"]), has_zero_point=c["zp"], qr=qr, round_delta=delta)
    want = ST[f"{name}|grad"]
    if oracle_dtype(c["dtype"]).__class__ is not str and oracle_dtype(c["dtype"]).is_exponent:
        np.testing.assert_allclose(grad, want, rtol=3e-7, atol=0)  # two float32 roundings (·ln2, /ln2) against torch's
    else:
        assert_bit_equal(grad, want, f"{name} gradient")
    assert (want == 0).any() or "qr" not in c  # the cases with a range do mask something
