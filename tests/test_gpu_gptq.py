"""GPTQ on the GPU (Hessian on MFMA, HIP column sweep) against the reference fixtures and the CPU oracle.

GPTQ is not bit-reproducible across BLAS / LAPACK implementations; the stated tolerance is: layer output
max-abs difference <= 1e-3 of the output's max-abs, and >= 99 % identical codes.  The Hessian itself is
checked against float64.
"""

import numpy as np
import pytest
import torch
from _product import build_quantizer, to_device, to_np
from conftest import load_golden
from test_oracle_gptq import dequant, layer_output_error

import gptq_oracle as go
import quant_oracle as qo
from deepcompressor_amd.data.cache import TensorCache
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.quantizer import QuantGptqConfig
from deepcompressor_amd.quantizer.kernel.gptq import cholesky_inverse_upper, gemm_f32, hessian_accumulate

pytestmark = pytest.mark.gpu
GQ = load_golden("gptq")
FQ = load_golden("fake_quant")


@pytest.mark.parametrize("T,C,dt", [(1024, 512, torch.bfloat16), (777, 200, torch.float16), (4096, 384, torch.bfloat16),
                                    (16384, 1024, torch.float16), (8192, 768, torch.bfloat16), (64, 256, torch.float16)])
def test_hessian_matches_float64(T, C, dt):
    """Whole 256-channel panels with whole 64-token tiles are read untransposed (transposing LDS reads); every other
    shape goes through the padded transposed copy."""
    g = torch.Generator().manual_seed(T + C)
    x = (torch.randn(T, C, generator=g) * torch.exp(torch.randn(C, generator=g))).to(dt)
    x[:, 3] = 0
    H = torch.zeros(C, C, device="cuda")
    hessian_accumulate(H, x.cuda(), 2.0 / T)
    hessian_accumulate(H, x.cuda(), 2.0 / T)  # accumulates
    want = (x.double().T @ x.double()) * (4.0 / T)
    got = H.cpu().double()
    assert torch.equal(got, got.T), "Hessian must be exactly symmetric"
    scale = want.diagonal().sqrt()
    rel = ((got - want).abs() / (scale[:, None] * scale[None, :] + 1e-30)).max().item()
    assert rel < 2e-6, rel
    assert got[3, 3] == 0


@pytest.mark.parametrize("T,C", [(16384, 4096), (8192, 2560), (4000, 1000)])
@pytest.mark.parametrize("width", [8, 100, 224])
def test_hessian_does_not_depend_on_the_round_width(T, C, width):
    """The round width of dcq_hessian_accumulate_ex only changes how many workgroups are launched at a time (direct and
    transposed-copy kernels, several K splits): the Hessian is the same bit for bit; the events it is handed bracket
    the launches."""
    g = torch.Generator().manual_seed(T + C)
    x = (torch.randn(T, C, generator=g) * torch.exp(torch.randn(C, generator=g))).to(torch.bfloat16).cuda()
    base = torch.zeros(C, C, device="cuda")
    hessian_accumulate(base, x, 2.0 / T)
    narrow = torch.zeros(C, C, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hessian_accumulate(narrow, x, 2.0 / T, round_width=width, start_event=e0, stop_event=e1)
    torch.cuda.synchronize()
    assert torch.equal(base, narrow)
    assert e0.elapsed_time(e1) > 0.0


def test_hessian_float32_activations():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1001, 264, generator=g) * torch.exp(torch.randn(264, generator=g))
    H = torch.zeros(264, 264, device="cuda")
    hessian_accumulate(H, x.cuda(), 2.0 / 1001)
    hessian_accumulate(H, x.cuda(), 2.0 / 1001)
    want = (x.double().T @ x.double()) * (4.0 / 1001)
    scale = want.diagonal().sqrt()
    rel = ((H.cpu().double() - want).abs() / (scale[:, None] * scale[None, :])).max().item()
    assert rel < 5e-6, rel


@pytest.mark.parametrize("M,N,K,a_t", [(128, 128, 32, False), (96, 200, 128, False), (300, 515, 64, True),
                                       (1024, 2048, 128, False), (257, 129, 256, True), (4, 4, 32, False), (131, 77, 64, True), (131, 77, 64, False),
                                       (2950, 3075, 64, False), (2950, 3076, 32, True), (3000, 2900, 160, False),
                                       (2901, 3000, 96, True)])
def test_gemm_f32_matches_float64(M, N, K, a_t):
    """c = beta c + alpha a b on the float32 matrix-core path: ragged M / N edges, both A layouts, strided views."""
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    b_full = torch.randn(K, N + 12 if N == 77 else (N + 12 + 3) // 4 * 4, generator=g)  # N = 77: unaligned load path
    c_full = torch.randn(M, N + 8, generator=g)
    b, c = b_full[:, 12:12 + N], c_full[:, 8:]  # column offsets keep 16-byte alignment but make the views strided
    want = 0.5 * c.double() - 1.25 * (a.double() @ b.double())
    a_dev = a.cuda()
    if a_t:
        a_dev = torch.zeros(K, M if N == 77 else (M + 3) // 4 * 4, device="cuda")[:, :M].copy_(a.T)
    b_dev, c_dev = b_full.cuda(), c_full.cuda()
    gemm_f32(a_dev, b_dev[:, 12:12 + N], c_dev[:, 8:], alpha=-1.25, beta=0.5, a_transposed=a_t)
    got = c_dev.cpu()
    assert torch.equal(got[:, :8], c_full[:, :8]), "columns outside the view must stay untouched"
    err = (got[:, 8:].double() - want).abs().max().item()
    assert err < 2e-5 * (K ** 0.5), err


@pytest.mark.parametrize("M,N,K,a_t", [(128, 128, 32, False), (96, 200, 128, False), (300, 516, 64, True),
                                       (1024, 2048, 128, False), (260, 132, 256, True), (4, 4, 32, False),
                                       (131, 76, 64, False), (2952, 3076, 32, True), (3001, 2900, 160, False),
                                       (2904, 3000, 96, True)])
def test_gemm_f32_two_term_matches_float64(M, N, K, a_t):
    """The two-float16-term path (per-row / per-column power-of-two scales, h h' + h l' + l h'): ragged edges, both A
    layouts, strided views, operands graded over twelve decades along M and N and four along K.  Bar: the error of
    every element within 8e-7 of sum |a||b| -- each operand's term pair carries a relative error of up to 2^-22 and the
    term left out another, 3 x 2.4e-7 at worst when K is too short to average them (observed: 4.7e-7 at K = 32, 2e-7 at
    K >= 512; a float32 multiply-accumulate chain in torch reaches 2.5e-7 ... 3.6e-7 on the long shapes)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    grade = lambda n: torch.logspace(-6, 6, n)[torch.randperm(n, generator=g)]
    a = torch.randn(M, K, generator=g) * grade(M)[:, None] * torch.logspace(-2, 2, K)[None, :]
    b_full = torch.randn(K, N + 12, generator=g)
    b_full[:, 12:] *= grade(N)[None, :]
    c_full = torch.randn(M, N + 8, generator=g)
    b, c = b_full[:, 12:12 + N], c_full[:, 8:]
    want = 0.5 * c.double() - 1.25 * (a.double() @ b.double())
    bound = 0.5 * c.double().abs() + 1.25 * (a.double().abs() @ b.double().abs())
    a_dev = a.T.contiguous().cuda() if a_t else a.cuda()
    b_dev, c_dev = b_full.cuda(), c_full.cuda()
    gemm_f32(a_dev, b_dev[:, 12:12 + N], c_dev[:, 8:], alpha=-1.25, beta=0.5, a_transposed=a_t, two_term=True)
    got = c_dev.cpu()
    assert torch.equal(got[:, :8], c_full[:, :8]), "columns outside the view must stay untouched"
    err = ((got[:, 8:].double() - want).abs() / bound).max().item()
    assert err < 8e-7, err


def test_gemm_f32_two_term_special_rows_and_refusals():
    """Rows / columns of zeros stay exact zeros, a non-finite value reaches exactly the results it belongs to, the
    smallest and largest float32 magnitudes survive the scaling; shapes and alignments the path does not take are
    refused, not computed some other way."""
    import ctypes as C

    from deepcompressor_amd import _lib

    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(64, 96, generator=g), torch.randn(96, 128, generator=g)
    a[3] = 0
    b[:, 5] = 0
    a[7] *= 1e-36
    b[:, 9] *= 1e36
    a[11, 2] = float("inf")
    b[4, 13] = float("nan")
    out = torch.full((64, 128), 7.0, device="cuda")
    gemm_f32(a.cuda(), b.cuda(), out, two_term=True)
    got, want = out.cpu(), a.double() @ b.double()
    assert (got[3] == 0)[torch.arange(128) != 13].all() and (got[:, 5] == 0)[torch.arange(64) != 11].all()
    bad = torch.zeros(64, 128, dtype=torch.bool)
    bad[11], bad[:, 13] = True, True
    assert not torch.isfinite(got[bad]).any() and torch.isfinite(got[~bad]).all()
    rel = ((got.double() - want).abs() / (a.double().abs() @ b.double().abs()).clamp_min(1e-300))[~bad]
    assert rel.max().item() < 8e-7
    lib = _lib.lib()
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    x = torch.zeros(64, 64, device="cuda")
    ptr = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
    call = lambda A, lda, kmajor, M, N, K, nbytes: lib.dcq_gemm_f32_two_term(A, lda, kmajor, ptr(x), 64, ptr(x), 64, M, N, K,
                                                                         1.0, 0.0, ptr(ws), nbytes, None)
    # dcq_status: DCQ_ERR_UNSUPPORTED = -2, DCQ_ERR_ARGUMENT = -1
    assert call(ptr(x), 64, 0, 64, 64, 48, 1 << 20) == -2       # K not a multiple of 32
    assert call(ptr(x), 64, 0, 64, 62, 32, 1 << 20) == -2       # N not a multiple of 4
    assert call(ptr(x), 64, 1, 62, 64, 32, 1 << 20) == -2       # k-major A: M not a multiple of 4
    assert call(ptr(x, 4), 64, 0, 32, 64, 32, 1 << 20) == -2    # A not 16-byte aligned
    assert call(ptr(x), 62, 0, 32, 64, 32, 1 << 20) == -2       # leading dimension not a multiple of 4
    assert call(ptr(x), 64, 0, 64, 64, 32, 256) == -1        # workspace below dcq_gemm_f32_two_term_workspace_bytes
    assert call(ptr(x), 64, 0, 63, 64, 32, 1 << 20) == 0                    # row-major A: any M
    # K = 0: an empty sum, C = beta * C
    c = torch.full((64, 64), 3.0, device="cuda")
    assert lib.dcq_gemm_f32_two_term(ptr(x), 64, 0, ptr(x), 64, ptr(c), 64, 64, 64, 0, 1.0, 0.5, ptr(ws), 1 << 20, None) == 0
    assert (c == 1.5).all()


@pytest.mark.parametrize("n,use_perm", [(128, False), (200, True), (640, True), (1000, False), (2048, True)])
def test_inverse_factor_matches_float64(n, use_perm):
    """U from the blocked factorisation + triangular inverse vs the reference route
    (cholesky -> cholesky_inverse -> cholesky(upper), gptq.py:215-233) evaluated in float64."""
    g = torch.Generator().manual_seed(n)
    x = torch.randn(4 * n, n, generator=g) * torch.exp(0.5 * torch.randn(n, generator=g))
    x[:, 5] = 0  # dead channel: zero row / column, diagonal treated as 1
    h = (x.T @ x) * (2.0 / x.shape[0])
    diag = torch.where(h.diagonal() == 0, torch.ones(n), h.diagonal())
    perm = torch.argsort(diag, descending=True) if use_perm else None
    mean = float(diag.mean())
    u, tries = cholesky_inverse_upper(h.cuda(), None if perm is None else perm.cuda(), mean, 0.01, 10)
    assert tries == 1
    s = h.double().clone()
    s.diagonal().copy_(diag.double())
    if perm is not None:
        s = s[perm][:, perm]
    s.diagonal().add_(0.01 * mean)
    want = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(s)), upper=True)
    got = u.cpu().double()
    assert torch.equal(got, got.triu()), "U must be upper triangular"
    rel = ((got - want).abs().max() / want.abs().max()).item()
    assert rel < 2e-4, rel
    resid = (got.T @ got @ s - torch.eye(n, dtype=torch.double)).abs().max().item()
    assert resid < 5e-3, resid


def test_inverse_factor_large_matches_float64():
    """n = 6016: the size from which the recursion's products run as three-term bf16 splits on the bf16 matrix pipe
    (triangular operands with clipped k ranges, the upper-only update, both A layouts).  Reference route in float64 on
    the GPU."""
    n = 6016
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(2 * n, n, generator=g, device="cuda") * torch.exp(0.5 * torch.randn(n, generator=g, device="cuda"))
    h = (x.T @ x) * (2.0 / x.shape[0])
    del x
    perm = torch.argsort(h.diagonal(), descending=True)
    mean = float(h.diagonal().mean())
    u, tries = cholesky_inverse_upper(h, perm, mean, 0.01, 10)
    assert tries == 1
    s = h.double()[perm][:, perm]
    s.diagonal().add_(0.01 * mean)
    want = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(s)), upper=True)
    got = u.double()
    assert torch.equal(got, got.triu()), "U must be upper triangular"
    rel = ((got - want).abs().max() / want.abs().max()).item()
    assert rel < 2e-4, rel
    resid = (got.T @ got @ s - torch.eye(n, dtype=torch.double, device="cuda")).abs().max().item()
    assert resid < 5e-3, resid


def test_inverse_factor_retries_with_more_damping():
    """A matrix that is not positive definite at the first damping level succeeds after retries (gptq.py:221-230)."""
    n = 256
    g = torch.Generator().manual_seed(1)
    q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.double))
    eig = torch.linspace(1.0, 2.0, n, dtype=torch.double)
    eig[0] = -0.0135  # needs damping > 0.0135: with mean(diag) ~ 1.5 and damp 0.005 that is the 9th try
    h = ((q * eig) @ q.T).float()
    h = (h + h.T) / 2
    mean = float(h.diagonal().mean())
    u, tries = cholesky_inverse_upper(h.cuda(), None, mean, 0.005, 50)
    assert 2 <= tries <= 30, tries
    s = h.double() + torch.eye(n, dtype=torch.double) * (0.005 * mean * (1.0 + 0.1 * (tries - 1)))
    assert torch.linalg.eigvalsh(s).min() > 0
    got = u.cpu().double()
    resid = (got.T @ got @ s - torch.eye(n, dtype=torch.double)).abs().max().item()
    assert resid < 5e-2, resid
    with pytest.raises(AssertionError):
        cholesky_inverse_upper(h.cuda(), None, mean, 0.005, 2)


def test_gptq_repeats_the_sweep_when_the_first_factorisation_fails():
    """``gptq_quantize`` queues the sweep behind the factorisation before it reads the factorisation's flag; when the
    flag reports a non-positive pivot, the factorisation and the sweep are repeated with more damping.  The result
    equals the one obtained with a factor that ``cholesky_inverse_upper`` (which waits for every attempt) found."""
    from deepcompressor_amd.data.dtype import QDType
    from deepcompressor_amd.data.zero import ZeroPointDomain
    from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod

    n, rows = 256, 64
    g = torch.Generator().manual_seed(1)
    q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.double))
    eig = torch.linspace(1.0, 2.0, n, dtype=torch.double)
    eig[0] = -0.0135
    h = ((q * eig) @ q.T).float()
    h = ((h + h.T) / 2).cuda().contiguous()
    w = (torch.randn(rows, n, generator=g) * 0.05).cuda()
    scale = (w.view(rows, 1, n // 128, 128).abs().amax(dim=-1, keepdim=True) / 7.5)
    zero = torch.full_like(scale, 8.0) * scale
    cfg = QuantGptqConfig(damp_percentage=0.005, block_size=128, num_inv_tries=50)
    calls = []
    plain = gptq_mod._enqueue_factor

    def counted(*args):
        calls.append(args[2])
        return plain(*args)

    gptq_mod._enqueue_factor = counted
    gptq_mod.clear_hessian_cache()
    try:
        codes = gptq_mod.gptq_quantize(w.clone(), view_shape=torch.Size((rows, 1, n // 128, 128)),
                                       quant_dtype=QDType.uint4, zero_domain=ZeroPointDomain.PostScale, scale=scale,
                                       zero=zero, gptq_config=cfg, inputs=None, hessian=h)
        first = len(calls)
        assert 2 <= first <= 30 and calls == sorted(calls), calls  # damping grew until the factorisation held
        # a second weight on the same Hessian reuses the verified factor: no further attempt
        again = gptq_mod.gptq_quantize(w.clone(), view_shape=torch.Size((rows, 1, n // 128, 128)),
                                       quant_dtype=QDType.uint4, zero_domain=ZeroPointDomain.PostScale, scale=scale,
                                       zero=zero, gptq_config=cfg, inputs=None, hessian=h)
        assert len(calls) == first and torch.equal(again, codes)
    finally:
        gptq_mod._enqueue_factor = plain
        gptq_mod.clear_hessian_cache()
    assert torch.isfinite(codes).all() and codes.min() >= 0 and codes.max() <= 15
    with pytest.raises(AssertionError):
        gptq_mod.gptq_quantize(w.clone(), view_shape=torch.Size((rows, 1, n // 128, 128)), quant_dtype=QDType.uint4,
                               zero_domain=ZeroPointDomain.PostScale, scale=scale, zero=zero,
                               gptq_config=QuantGptqConfig(damp_percentage=0.005, num_inv_tries=2), inputs=None,
                               hessian=h)
    gptq_mod.clear_hessian_cache()


@pytest.mark.parametrize("tag", sorted(GQ.meta))
def test_gptq_matches_reference_fixture(tag):
    m = GQ.meta[tag]
    case = FQ.meta["cases"][m["case"]]
    w_np, x_np = GQ[f"{tag}|w"], GQ[f"{tag}|x"]
    w = to_device(w_np, m["in_dtype"])
    x = to_device(x_np, m["in_dtype"], device="cpu")  # the calibration cache lives on the host, as in the reference
    q = build_quantizer(case)
    cache = TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_cached=1, num_total=1,
                        num_samples=m["num_samples"])
    kern = QuantGptqConfig(damp_percentage=m["damp"], block_size=m["block"], num_inv_tries=250, hessian_block_size=512)
    qt = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True)
    dq = q.quantize(w, kernel=kern, inputs=cache)
    np.testing.assert_array_equal(to_np(qt.scale.data), GQ[f"{tag}|scale"])  # scale/zero: bit-exact (pre-update weights)
    codes, want = to_np(qt.qdata), GQ[f"{tag}|codes"]
    same = (codes == want).mean()
    assert same > 0.99, f"only {same:.4f} of the codes equal the reference's"
    err = layer_output_error(x_np, to_np(dq.data), GQ[f"{tag}|dequant"])
    assert err < 1e-3, f"layer output differs from the reference's by {err:.2e}"
    e_gptq = layer_output_error(x_np, to_np(dq.data), w_np)
    e_rtn = layer_output_error(x_np, GQ[f"{tag}|rtn_dequant"], w_np)
    assert e_gptq < e_rtn
    # the dequantized tensor is exactly the dequantization of the returned codes
    spec = qo.spec_from_case(case)
    base = qo.fake_quantize(w_np, m["in_dtype"], spec)
    again = qo.round_to(m["in_dtype"], dequant(codes, base, spec.steps[0], w_np.shape))
    np.testing.assert_array_equal(to_np(dq.data), again)


def test_gptq_vs_oracle_larger():
    """768 x 1024 weight, 4096 tokens, ragged last sweep block excluded (C multiple of 128), dead channel included."""
    case = FQ.meta["cases"]["uint4_post_g128_f16s"]
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(768, 1024, generator=g) * 0.02).to(torch.bfloat16)
    x = (torch.randn(4096, 1024, generator=g) * torch.exp(0.7 * torch.randn(1024, generator=g))).to(torch.bfloat16)
    x[:, 100] = 0
    spec = qo.spec_from_case(case)
    base = qo.fake_quantize(w.float().numpy(), "bf16", spec)
    want = go.gptq_codes(w.float().numpy(), x.float().numpy(), 4096, spec.steps[0], base["scale"], base["zero"],
                         base["view_shape"])
    q = build_quantizer(case)
    cache = TensorCache([x.view(2, 2048, 1024)], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=4096)
    qt = q.quantize(w.cuda(), kernel=QuantGptqConfig(hessian_block_size=512), inputs=cache, return_with_dequant=False,
                    return_with_quant=True)
    codes = to_np(qt.qdata)
    assert (codes == want).mean() > 0.99
    err = layer_output_error(x.float().numpy(), dequant(codes, base, spec.steps[0], w.shape),
                             dequant(want, base, spec.steps[0], w.shape))
    assert err < 1e-3, err
    assert (codes[:, 100] == want[:, 100]).all()


def test_gptq_ragged_columns_and_progressive():
    """C = 320 (last sweep block has 64 columns) and the QoQ two-step config through the GPTQ kernel."""
    g = torch.Generator().manual_seed(3)
    for cname, C in (("sint4_g64_defs", 320), ("qoq_g128", 384)):
        case = FQ.meta["cases"][cname]
        w = (torch.randn(96, C, generator=g) * 0.03).to(torch.float16)
        x = (torch.randn(1024, C, generator=g)).to(torch.float16)
        q = build_quantizer(case)
        cache = TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=1024)
        dq = q.quantize(w.cuda(), kernel=QuantGptqConfig(), inputs=cache).data
        rtn = q.quantize(w.cuda()).data
        xn, wn = x.float().numpy(), w.float().numpy()
        e_gptq = np.square(xn.astype(np.float64) @ (to_np(dq) - wn).T.astype(np.float64)).sum()
        e_rtn = np.square(xn.astype(np.float64) @ (to_np(rtn) - wn).T.astype(np.float64)).sum()
        assert e_gptq < e_rtn, (cname, e_gptq, e_rtn)


def test_gptq_far_updates_with_and_without_scratch(monkeypatch):
    """dcq_gptq_sweep with its optional workspace (far updates as two-term products) and without it (three-term
    products): two float32-accurate roundings of the same sums, so the codes agree but for the rare weight that sits
    on a rounding boundary, and the layer outputs agree to the GPTQ bar."""
    from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod

    g = torch.Generator().manual_seed(77)
    rows, cols, tokens = 96, 4096, 4608
    w = (torch.randn(rows, cols, generator=g) * 0.02).to(torch.bfloat16).cuda()
    x = (torch.randn(tokens, cols, generator=g) * torch.exp(0.5 * torch.randn(cols, generator=g))).to(torch.bfloat16).cuda()
    q = build_quantizer(FQ.meta["cases"]["uint4_post_g128_f16s"])
    kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
    cache = TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_cached=1, num_total=1, num_samples=tokens)
    outs = []
    for two_term in (True, False):
        monkeypatch.setattr(gptq_mod, "TWO_TERM_UPDATES", two_term)
        outs.append(q.quantize(w, kernel=kern, inputs=cache, return_with_quant=True))
    same = (outs[0].qdata == outs[1].qdata).float().mean().item()
    assert same >= 0.999, same
    y0, y1 = x.float() @ outs[0].data.float().T, x.float() @ outs[1].data.float().T
    assert ((y0 - y1).abs().max() / y0.abs().max()).item() <= 1e-3


@pytest.mark.parametrize("rows,cols,tokens", [(640, 768, 3000), (384, 3072, 4096)])
def test_gptq_row_shards_equal_unsharded(rows, cols, tokens):
    """Rows are independent given the shared Hessian: quantizing row slabs separately (what each rank does when one
    linear is sharded over GPUs) reproduces the unsharded codes bit for bit.  The wide case reaches the lazy updates
    that run as bf16 splits: which arithmetic an update uses is decided by its column count, never by the rows."""
    from deepcompressor_amd.parallel import row_shard

    case = FQ.meta["cases"]["uint4_post_g128_f16s"]
    g = torch.Generator().manual_seed(21)
    w = (torch.randn(rows, cols, generator=g) * 0.02).to(torch.bfloat16).cuda()
    x = (torch.randn(tokens, cols, generator=g) * torch.exp(0.5 * torch.randn(cols, generator=g))).to(torch.bfloat16).cuda()
    cache = TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=tokens, orig_device=x.device)
    q = build_quantizer(case)
    kern = QuantGptqConfig()
    full = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True).qdata
    world = 3
    slabs = []
    for rank in range(world):
        b, e = row_shard(w.shape[0], world, rank)
        slabs.append(q.quantize(w[b:e].contiguous(), kernel=kern, inputs=cache, return_with_dequant=False,
                                return_with_quant=True).qdata)
    assert torch.equal(torch.cat(slabs, dim=0), full)


def test_groups_on_streams_equal_serial():
    """Independent linears run side by side (one HIP stream and host thread each, Hessians prepared beforehand)
    give bit-identical codes to running them one after the other."""
    from deepcompressor_amd.parallel import run_on_streams
    from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod

    case = FQ.meta["cases"]["uint4_post_g128_f16s"]
    g = torch.Generator().manual_seed(5)
    shapes = [(256, 512), (384, 768), (128, 1024), (512, 256)]
    ws = [(torch.randn(r, c, generator=g) * 0.02).to(torch.bfloat16).cuda() for r, c in shapes]
    xs = [(torch.randn(2048, c, generator=g) * torch.exp(0.5 * torch.randn(c, generator=g))).to(torch.bfloat16).cuda()
          for _, c in shapes]
    caches = [TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_samples=2048, orig_device=x.device)
              for x in xs]
    kern = QuantGptqConfig()

    def job(i):
        q = build_quantizer(case)
        return q.quantize(ws[i], kernel=kern, inputs=caches[i], return_with_dequant=False, return_with_quant=True).qdata

    serial = [job(i) for i in range(len(ws))]
    gptq_mod.clear_hessian_cache()
    for _ in range(2):
        hs = gptq_mod.prepare_hessians(caches, ws[0].device)
        assert len(hs) == len(ws) and gptq_mod.hessian_of(caches[2], ws[0].device) is hs[2]
        side = run_on_streams([lambda i=i: job(i) for i in range(len(ws))])
        gptq_mod.clear_hessian_cache()
        for a, b in zip(serial, side):
            assert torch.equal(a, b)


def test_run_on_streams_propagates_errors():
    from deepcompressor_amd.parallel import run_on_streams

    def bad():
        raise ValueError("boom")

    with pytest.raises(ValueError, match="boom"):
        run_on_streams([lambda: torch.ones(4, device="cuda"), bad])


@pytest.mark.parametrize("rows,cols", [(64, 512), (33, 14336), (5, 36864), (7, 130), (16, 40960)])
def test_gather_columns_equals_the_indexing_expression(rows, cols):
    """The act-order gathers around the sweep (W[:, perm] with dead columns zeroed, Q[:, inverse_perm]) through the
    LDS-staged kernel, and through the torch expression for the shapes the kernel does not take."""
    from deepcompressor_amd import engine
    g = torch.Generator().manual_seed(rows * 131 + cols)
    src = torch.randn(rows, cols, generator=g).cuda()
    perm = torch.randperm(cols, generator=g).cuda()
    dead = (torch.rand(cols, generator=g) < 0.05).cuda()
    want = torch.where(dead, torch.zeros((), device="cuda"), src)[:, perm]
    got = engine.gather_columns(src, perm, zero_columns=dead)
    assert got.is_contiguous() and torch.equal(got, want)
    inverse = torch.argsort(perm)
    assert torch.equal(engine.gather_columns(got, inverse), want[:, inverse])
    assert torch.equal(engine.gather_columns(src, inverse), src[:, inverse])
    for dt in (torch.bfloat16, torch.float16):  # 16-bit sources are widened on the way
        narrow = src.to(dt)
        want16 = torch.where(dead, torch.zeros((), device="cuda"), narrow.float())[:, perm]
        got16 = engine.gather_columns(narrow, perm, zero_columns=dead)
        assert got16.dtype == torch.float32 and torch.equal(got16, want16)


@pytest.mark.parametrize("n", [1, 3, 4, 1000003, 1 << 22])
def test_assert_finite_names_what_it_found(n):
    """The NaN / Inf assertions of the reference (impl/base.py:250-251, gptq.py:229-230) through the one-pass kernel:
    silent on finite data, 'contains NaN.' / 'contains Inf.' in the reference's order otherwise."""
    from deepcompressor_amd import engine
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g).cuda()
    engine.assert_finite(x, "Quantized tensor")
    for pos in sorted({0, n // 2, n - 1}):
        for bad, name in ((float("nan"), "NaN"), (float("inf"), "Inf"), (float("-inf"), "Inf")):
            y = x.clone()
            y[pos] = bad
            with pytest.raises(AssertionError, match=f"Quantized tensor contains {name}."):
                engine.assert_finite(y, "Quantized tensor")
    if n >= 3:
        y = x.clone()
        y[0], y[n - 1] = float("inf"), float("nan")
        with pytest.raises(AssertionError, match="contains NaN."):
            engine.assert_finite(y, "Quantized tensor")
        with pytest.raises(AssertionError, match="contains Inf."):
            engine.assert_finite(y, "Quantized tensor", inf_first=True)


def test_gather_and_finite_check_on_empty_inputs():
    from deepcompressor_amd import engine
    empty = torch.empty(0, 256, device="cuda")
    out = engine.gather_columns(empty, torch.randperm(256).cuda())
    assert out.shape == (0, 256) and out.dtype == torch.float32
    engine.assert_finite(torch.empty(0, device="cuda"), "Quantized tensor")


GC = load_golden("gptq_conv")


@pytest.mark.parametrize("tag", sorted(GC.meta))
def test_gptq_on_a_convolution_weight_matches_reference_fixture(tag):
    """Views with more than two pairs of dimensions (a convolution's (out, in, kh, kw) weight): the columns of a
    quantization group are moved side by side for the sweep and back afterwards as the reference does
    (quantizer/kernel/gptq.py:168-178, 272-281), the Hessian comes from the unfolded inputs."""
    from deepcompressor_amd.data.utils.reshape import ConvInputReshapeFn
    m = GC.meta[tag]
    w_np, x_np = GC[f"{tag}|w"], GC[f"{tag}|x"]
    w, x = to_device(w_np, "bf16"), to_device(x_np, "bf16")
    q = build_quantizer(m["case"])
    cache = TensorCache([x], channels_dim=1, reshape=ConvInputReshapeFn((3, 3), (1, 1), (1, 1), (1, 1)), num_cached=1,
                        num_total=1, num_samples=m["num_samples"])
    kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
    qt = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True)
    dq = q.quantize(w, kernel=kern, inputs=cache)
    assert list(qt.view_shape) == m["view_shape"]
    np.testing.assert_array_equal(to_np(qt.scale.data), GC[f"{tag}|scale"])
    np.testing.assert_array_equal(to_np(qt.zero), GC[f"{tag}|zero"])
    codes, want = to_np(qt.qdata), GC[f"{tag}|codes"]
    assert codes.shape == want.shape
    same = (codes == want).mean()
    assert same > 0.99, f"only {same:.4f} of the codes equal the reference's"
    y = lambda wt: torch.nn.functional.conv2d(torch.from_numpy(x_np).double(), torch.from_numpy(wt).double(), padding=1)  # noqa: E731
    err = ((y(to_np(dq.data)) - y(GC[f"{tag}|dequant"])).abs().max() / y(GC[f"{tag}|dequant"]).abs().max()).item()
    assert err < 1e-3, err
