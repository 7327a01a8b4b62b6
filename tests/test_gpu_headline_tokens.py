"""The headline workload at its real calibration size: 128 x 2048 = 262,144 tokens.

At that size ``dcq_hessian_accumulate`` runs plans no small test reaches (C = 4096: many-way split of the token
range with a fixed-order reduction of partial tiles; C = 14336: few splits over thousands of k-tiles).  Each plan is
checked (a) against the same activations fed as 16 chunks of 16,384 tokens, the size the small tests verify against
float64, and (b) on a 512-column slab against a float64 product; then GPTQ on k_proj (1024 x 4096) with all 262,144
tokens is compared on a 16-row slab with the CPU oracle, whose Hessian is the reference's 512-row block loop
(deepcompressor/quantizer/kernel/gptq.py:183-201).
"""

import numpy as np
import pytest
import torch
from _product import build_quantizer, to_np
from conftest import load_golden
from test_oracle_gptq import dequant, layer_output_error

import gptq_oracle as go
import quant_oracle as qo
from deepcompressor_amd.data.cache import TensorCache
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.quantizer import QuantGptqConfig
from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod

pytestmark = pytest.mark.gpu
FQ = load_golden("fake_quant")
TOKENS = 128 * 2048


def activations(channels: int, seed: int) -> torch.Tensor:
    """bf16 [262144][channels] with per-channel log-normal gains, generated on the GPU in 16,384-token pieces."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    gains = torch.exp(torch.randn(channels, generator=g, device=dev))
    x = torch.empty((TOKENS, channels), dtype=torch.bfloat16, device=dev)
    for t0 in range(0, TOKENS, 16384):
        x[t0:t0 + 16384] = (torch.randn((16384, channels), generator=g, device=dev) * gains).to(torch.bfloat16)
    return x


@pytest.mark.parametrize("channels", [4096, 14336])
def test_hessian_at_262144_tokens(channels):
    x = activations(channels, channels)
    alpha = 2.0 / TOKENS
    whole = torch.zeros(channels, channels, device=x.device)
    gptq_mod.hessian_accumulate(whole, x, alpha)
    assert torch.equal(whole, whole.T), "Hessian must be exactly symmetric"
    # (a) the same tokens through the plan of the 16,384-token launches
    pieces = torch.zeros_like(whole)
    for t0 in range(0, TOKENS, 16384):
        gptq_mod.hessian_accumulate(pieces, x[t0:t0 + 16384], alpha)
    norm = whole.diagonal().sqrt()
    rel = ((whole - pieces).abs() / (norm[:, None] * norm[None, :])).max().item()
    # float32 accumulation over 262,144 tokens in two different orders (measured: 1.2e-6 / 4.2e-6 for C = 4096 /
    # 14336, whose plan adds up to 87,000 tokens in one chain); the reference's own float32 sum over 512 blocks of 512
    # rows is no closer to the exact value
    assert rel < 1e-5, rel
    print(f"C={channels}: whole vs 16 pieces {rel:.3g}")
    del pieces
    # (b) a 512-column slab in float64 (columns chosen to straddle a tile boundary and end at the last channel)
    for c0 in (192, channels - 512):
        want = torch.zeros(channels, 512, dtype=torch.float64, device=x.device)
        for t0 in range(0, TOKENS, 32768):
            xb = x[t0:t0 + 32768].double()
            want += xb.T @ xb[:, c0:c0 + 512]
        want *= alpha
        rel = ((whole[:, c0:c0 + 512].double() - want).abs()
               / (norm[:, None].double() * norm[None, c0:c0 + 512].double())).max().item()
        assert rel < 1e-5, (c0, rel)
        print(f"C={channels}: columns [{c0}, {c0 + 512}) vs float64 {rel:.3g}")
    gptq_mod.release_workspaces()


def test_gptq_k_proj_at_262144_tokens_against_the_oracle():
    """GPTQ W4 g128 (uint4 + PostScale zero, fp16 scales) on k_proj with the full calibration set; a 16-row slab of the
    result against the CPU oracle.  Bars: >= 99 % identical codes, layer output within 1e-3 (relative to its max-abs)."""
    case = FQ.meta["cases"]["uint4_post_g128_f16s"]
    spec = qo.spec_from_case(case)
    rows, cols = 1024, 4096
    x = activations(cols, 7)
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(rows, cols, generator=g) * 0.02).to(torch.bfloat16)
    cache = TensorCache([x.view(128, 2048, cols)], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=TOKENS,
                        orig_device=x.device)
    q = build_quantizer(case)
    kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
    gptq_mod.clear_hessian_cache()
    try:
        codes = q.quantize(w.cuda(), kernel=kern, inputs=cache, return_with_dequant=False,
                           return_with_quant=True).qdata
    finally:
        gptq_mod.clear_hessian_cache()
    assert codes.min() >= 0 and codes.max() <= 15 and torch.equal(codes, codes.round())
    slab = slice(500, 516)
    w_np = w[slab].float().numpy()
    x_np = x.float().cpu().numpy()
    base = qo.fake_quantize(w_np, "bf16", spec)
    want = go.gptq_codes(w_np, x_np, TOKENS, spec.steps[0], base["scale"], base["zero"], base["view_shape"])
    got = to_np(codes[slab])
    assert (got == want).mean() > 0.99, float((got == want).mean())
    err = layer_output_error(x_np[:8192], dequant(got, base, spec.steps[0], w_np.shape),
                             dequant(want, base, spec.steps[0], w_np.shape))
    assert err < 1e-3, err


def test_inverse_factor_at_14336_matches_float64():
    """down_proj's inverse factor at its real order (n = 14336: the recursion's largest products on the three-term
    bf16 split path) against the reference route cholesky -> cholesky_inverse -> cholesky(upper)
    (deepcompressor/quantizer/kernel/gptq.py:215-233) evaluated in float64 on the GPU."""
    n = 14336
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n)
    gains = torch.exp(torch.randn(n, generator=g, device=dev))
    h = torch.zeros(n, n, device=dev)
    tokens = 2 * n
    for t0 in range(0, tokens, 4096):  # bf16 activations with log-normal gains, as the bench feeds them
        x = (torch.randn(4096, n, generator=g, device=dev) * gains).to(torch.bfloat16)
        gptq_mod.hessian_accumulate(h, x, 2.0 / tokens)
    del x
    perm = torch.argsort(h.diagonal(), descending=True)
    mean = float(h.diagonal().mean())
    u, tries = gptq_mod.cholesky_inverse_upper(h, perm, mean, 0.01, 10)
    assert tries == 1
    assert torch.equal(u, u.triu()), "U must be upper triangular"
    s = h.double()[perm][:, perm]
    del h
    s.diagonal().add_(0.01 * mean)
    want = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(s)), upper=True)
    got = u.double()
    del u
    rel = ((got - want).abs().max() / want.abs().max()).item()
    # relative to each column's own scale (the sweep divides by U[c][c] and multiplies by U[c][c:])
    col = ((got - want).abs().amax(dim=0) / want.abs().amax(dim=0)).max().item()
    del want
    print(f"n={n}: max |U - U64| / max |U64| = {rel:.3g}, per column {col:.3g}")
    assert rel < 2e-4, rel
    prod = got.T @ got
    resid = (prod @ s - torch.eye(n, dtype=torch.double, device=dev)).abs().max().item()
    print(f"n={n}: max |U^T U S - I| = {resid:.3g}")
    assert resid < 5e-3, resid
    gptq_mod.release_workspaces()


def _gptq_full_height_vs_oracle(rows: int, cols: int, seed: int, slab: slice, shard: slice | None):
    """GPTQ W4 g128 on a rows x cols weight with all 262,144 tokens: the full-height result on the GPU; ``slab`` (16
    rows of it) against the CPU oracle fed the GPU's Hessian (the Hessian itself is checked above); ``shard``: those
    rows quantized on their own must equal the same rows of the full-height run bit for bit (the row count picks the
    sweep kernel's launch shape, the result may not depend on it)."""
    case = FQ.meta["cases"]["uint4_post_g128_f16s"]
    spec = qo.spec_from_case(case)
    x = activations(cols, seed)
    g = torch.Generator().manual_seed(seed + 1)
    w = (torch.randn(rows, cols, generator=g) * 0.02).to(torch.bfloat16)
    cache = TensorCache([x.view(128, 2048, cols)], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=TOKENS,
                        orig_device=x.device)
    q = build_quantizer(case)
    kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
    gptq_mod.clear_hessian_cache()
    try:
        wd = w.cuda()
        codes = q.quantize(wd, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True).qdata
        h_np = gptq_mod.hessian_of(cache, x.device).cpu().numpy()
        if shard is not None:
            part = q.quantize(wd[shard].contiguous(), kernel=kern, inputs=cache, return_with_dequant=False,
                              return_with_quant=True).qdata
            assert torch.equal(part, codes[shard]), "a shard of the rows must give the rows of the whole"
    finally:
        gptq_mod.clear_hessian_cache()
    assert codes.shape == (rows, cols)
    assert codes.min() >= 0 and codes.max() <= 15 and torch.equal(codes, codes.round())
    w_np = w[slab].float().numpy()
    base = qo.fake_quantize(w_np, "bf16", spec)
    want = go.gptq_codes(w_np, None, TOKENS, spec.steps[0], base["scale"], base["zero"], base["view_shape"], H=h_np)
    got = to_np(codes[slab])
    same = float((got == want).mean())
    x_np = x[:8192].float().cpu().numpy()
    err = layer_output_error(x_np, dequant(got, base, spec.steps[0], w_np.shape),
                             dequant(want, base, spec.steps[0], w_np.shape))
    # and GPTQ did its job: a smaller output error than round-to-nearest with the same scales
    rtn = to_np(q.quantize(w[slab].cuda(), return_with_dequant=False, return_with_quant=True).qdata)
    e_gptq = layer_output_error(x_np, dequant(got, base, spec.steps[0], w_np.shape), w_np)
    e_rtn = layer_output_error(x_np, dequant(rtn, base, spec.steps[0], w_np.shape), w_np)
    print(f"{rows}x{cols}: identical codes {same:.4f}, layer output vs oracle {err:.3g}, "
          f"error vs the weight: GPTQ {e_gptq:.3g} / RTN {e_rtn:.3g}")
    assert same > 0.99, same
    assert err < 1e-3, err
    assert e_gptq < e_rtn
    gptq_mod.release_workspaces()


def test_gptq_gate_proj_at_262144_tokens_against_the_oracle():
    """gate_proj / up_proj (14336 x 4096): the sweep's launch shape for 14336 rows, a slab of rows that straddles
    workgroups, and a 1792-row shard (an eighth of the rows, as eight ranks would split them)."""
    _gptq_full_height_vs_oracle(14336, 4096, 11, slice(7160, 7176), slice(3 * 1792, 4 * 1792))


def test_gptq_down_proj_at_262144_tokens_against_the_oracle():
    """down_proj (4096 x 14336): 14336-column Hessian from the transposed-copy SYRK plan, the n = 14336 inverse
    factor, 112 sweep blocks with their lazy updates."""
    _gptq_full_height_vs_oracle(4096, 14336, 13, slice(2040, 2056), slice(512, 1024))


def test_hessian_memo_follows_the_tensors_not_their_addresses():
    """A freed cache whose memory the allocator hands to the next cache of the same shape must not be mistaken for it."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    gptq_mod.clear_hessian_cache()
    try:
        x1 = torch.randn((2048, 256), generator=g, device=dev).to(torch.bfloat16)
        c1 = TensorCache([x1], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=2048, orig_device=dev)
        with torch.inference_mode():
            h1 = gptq_mod.hessian_of(c1, dev).clone()
            del c1, x1
            x2 = torch.randn((2048, 256), generator=g, device=dev).to(torch.bfloat16) * 3
            c2 = TensorCache([x2], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=2048, orig_device=dev)
            h2 = gptq_mod.hessian_of(c2, dev)
        want = (x2.double().T @ x2.double()) * (2.0 / 2048)
        assert torch.allclose(h2.double(), want, rtol=1e-4, atol=1e-4), "a stale Hessian was returned"
        assert not torch.equal(h1, h2)
        assert gptq_mod.hessian_of(c2, dev) is h2  # the same cache object still hits the memo
    finally:
        gptq_mod.clear_hessian_cache()
