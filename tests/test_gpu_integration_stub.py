"""The reference-side binding printed in the repository's integration notes, executed as printed.

The block that replaces ``deepcompressor/csrc/load.py`` (a ctypes ``_C`` with ``round_to_nearest_in_codebook_cuda``) is
cut out of the document, run with the library's real path, and called the way ``Codebook.round`` calls it
(deepcompressor/data/codebook.py:37-50): in place and out of place, float32 / float16 / bfloat16, on the codebooks
captured from the reference (tests/golden/tables.npz) with every midpoint (tie -> upper value) in the inputs."""

import os
import re

import numpy as np
import pytest
import torch
from conftest import ROOT, assert_bit_equal, load_golden

import quant_oracle as qo

pytestmark = pytest.mark.gpu
TABLES = load_golden("tables")


def printed_stub() -> str:
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## The reference-side binding a maintainer would add"):]
    block = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    assert "round_to_nearest_in_codebook_cuda" in block and 'CDLL("libdcq.so")' in block
    return block.replace('CDLL("libdcq.so")', f'CDLL({os.path.join(ROOT, "deepcompressor_amd", "libdcq.so")!r})')


@pytest.mark.parametrize("name", ["sfp4_e2m1_all", "sfp8_e4m3_nan", "sfp6_e3m2_all"])
@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
def test_printed_ctypes_stub_rounds_like_the_reference_kernel(name, dt):
    scope: dict = {}
    exec(printed_stub(), scope)  # noqa: S102 -- the document's own code block
    _C = scope["_C"]
    cb = TABLES[f"codebook|{name}"].astype(np.float32)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50_001, generator=g) * float(cb.max()) / 2
    x = torch.cat([x, torch.from_numpy((cb[:-1] + cb[1:]) / 2), torch.from_numpy(cb)]).to(dt)
    want = qo.codebook_round(x.float().numpy(), cb)  # values of the codebook are exact in every 16-bit type used here
    xd, cbd = x.cuda(), torch.from_numpy(cb).cuda()
    out = _C.round_to_nearest_in_codebook_cuda(xd, cbd, False, False)
    assert out.data_ptr() != xd.data_ptr() and out.dtype == dt and torch.equal(xd.cpu(), x), "out of place"
    assert_bit_equal(out.float().cpu().numpy(), want, f"{name} out of place")
    same = _C.round_to_nearest_in_codebook_cuda(xd, cbd, True, False)
    assert same.data_ptr() == xd.data_ptr(), "in place"
    assert_bit_equal(xd.float().cpu().numpy(), want, f"{name} in place")


@pytest.mark.parametrize("name", ["sfp4_e2m1_all", "sfp8_e4m3_nan"])
def test_printed_ctypes_stub_takes_float64_tensors(name):
    """The reference's kernel is instantiated for float and double (csrc/quantize/quantize.cu:145) and compares in the
    tensor's type: float64 inputs that differ from a midpoint only beyond float32 precision must fall on the side the
    float64 comparison puts them, which a float32 kernel cannot tell."""
    scope: dict = {}
    exec(printed_stub(), scope)  # noqa: S102 -- the document's own code block
    _C = scope["_C"]
    cb = TABLES[f"codebook|{name}"].astype(np.float64)
    g = torch.Generator().manual_seed(4)
    mids = (cb[:-1] + cb[1:]) / 2
    mids = mids[cb[1:] > cb[:-1]]
    x = np.concatenate([torch.randn(20_001, generator=g, dtype=torch.float64).numpy() * cb.max() / 2, mids, cb,
                        mids * (1 - 1e-12), mids * (1 + 1e-12), mids - 1e-13, mids + 1e-13])
    lo = np.maximum(np.searchsorted(cb, x, side="right") - 1, 0)  # quantize.cu:69-99 in float64
    hi = np.minimum(lo + 1, len(cb) - 1)
    want = np.where(x + x < cb[lo] + cb[hi], cb[lo], cb[hi])
    xd = torch.from_numpy(x).cuda()
    out = _C.round_to_nearest_in_codebook_cuda(xd, torch.from_numpy(cb).cuda(), False, False)
    assert out.dtype == torch.float64
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    from deepcompressor_amd.data.dtype import QuantDataType
    book = QuantDataType.from_str(name).get_codebook(device="cuda")
    np.testing.assert_array_equal(book.round(xd).cpu().numpy(), want)
