"""Rotation on the GPU (float64 matrix-core GEMM) against the reference's fixtures and the CPU oracle.

Bar: bit-exact on the fixtures.  The float64 sums run in a different order than the reference's BLAS, so an
element can differ only where the float64 result sits within ~1e-15 (relative) of a rounding boundary of the
output dtype; on the larger random cases at most 1 element in 10^5 may differ, by one unit in the last place.
"""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import to_device, to_np
from conftest import assert_bit_equal, load_golden

import quant_oracle as qo
import rotate_oracle as ro
from deepcompressor_amd.calib.rotate import (get_rotation_matrix, hadamard_in_channels, rotate_in_channels,
                                             rotate_out_channels, transform_rms_norm_and_linear)
from deepcompressor_amd.utils.math.hadamard import HadamardMatrix, gemm_f64, hardmard_transform

pytestmark = pytest.mark.gpu
G = load_golden("rotate")
CASES = [k for k, v in G.meta.items() if "n" in v]
OUT_CASES = [k for k, v in G.meta.items() if "heads" in v]
TORCH = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


@pytest.mark.parametrize("M,N,K,batch,a_t,dts", [
    (64, 64, 16, 1, False, ("f64", "f64", "f64")), (70, 45, 28, 3, False, ("bf16", "f64", "bf16")),
    (33, 130, 100, 2, True, ("f64", "f16", "f16")), (128, 256, 512, 1, False, ("f32", "f64", "f32")),
    (5, 7, 3, 4, True, ("f64", "bf16", "f64")), (257, 129, 64, 1, False, ("f16", "f64", "f64")),
])
def test_gemm_f64_matches_numpy(M, N, K, batch, a_t, dts):
    tt = dict(TORCH, f64=torch.float64)
    g = torch.Generator().manual_seed(M * 31 + N)
    a = torch.randn(batch, M, K, generator=g, dtype=torch.float64).to(tt[dts[0]])
    b = torch.randn(K, N, generator=g, dtype=torch.float64).to(tt[dts[1]])  # shared by the batch
    want64 = a.double().numpy() @ b.double().numpy()
    a_dev = a.transpose(1, 2).contiguous().cuda() if a_t else a.cuda()
    got = gemm_f64(a_dev, b.cuda(), tt[dts[2]], a_transposed=a_t).cpu()
    assert got.shape == (batch, M, N)
    if dts[2] == "f64":
        assert np.abs(got.numpy() - want64).max() <= 1e-13 * K
    else:
        want = torch.from_numpy(want64).to(tt[dts[2]])
        assert (got != want).float().mean().item() <= 1e-4
        assert np.abs(got.double().numpy() - want64).max() <= np.abs(want64).max() * 2.0 ** -7


@pytest.mark.parametrize("tag", CASES)
def test_rotate_in_channels_fixture(tag):
    m = G.meta[tag]
    for name in ("fixed", "random"):
        p = nn.Parameter(to_device(G[f"{tag}|w"], m["dtype"]), requires_grad=False)
        rotate_in_channels(p, rotation=torch.from_numpy(G[f"{tag}|rotation_{name}"]))
        assert p.dtype == TORCH[m["dtype"]] and p.shape == (m["rows"], m["n"])
        assert_bit_equal(to_np(p.data), G[f"{tag}|in_{name}"], f"{tag} in_{name}")


@pytest.mark.parametrize("tag", OUT_CASES)
def test_rotate_out_channels_fixture(tag):
    m = G.meta[tag]
    pw = nn.Parameter(to_device(G[f"{tag}|w"], m["dtype"]), requires_grad=False)
    pb = nn.Parameter(to_device(G[f"{tag}|b"], m["dtype"]), requires_grad=False)
    rotate_out_channels(pw, rotation=torch.from_numpy(G[f"{tag}|rotation"]), bias=pb)
    assert_bit_equal(to_np(pw.data), G[f"{tag}|w_out"], f"{tag} w")
    assert_bit_equal(to_np(pb.data), G[f"{tag}|b_out"], f"{tag} b")


@pytest.mark.parametrize("tag", CASES)
def test_hadamard_in_channels_fixture(tag):
    m = G.meta[tag]
    lin = nn.Linear(m["n"], m["rows"], bias=False, device="cuda", dtype=TORCH[m["dtype"]])
    lin.weight.data = to_device(G[f"{tag}|w"], m["dtype"])
    hadamard_in_channels([lin])
    assert_bit_equal(to_np(lin.weight.data), G[f"{tag}|hadamard_in"], f"{tag} hadamard_in")
    # the registered hook applies the same transform to the inputs: H is orthogonal, so x H (W H)^T = x W^T
    x = torch.randn(9, m["n"], device="cuda", dtype=torch.float32).to(TORCH[m["dtype"]])
    y = lin(x)
    w0 = to_device(G[f"{tag}|w"], m["dtype"]).float()
    ref = x.float() @ w0.T
    assert (y.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item() + 1e-3
    # float64 transform with unscaled matrices (the 1 / sqrt(n) multiply happens in hardmard_transform)
    rhs, lhs, k = HadamardMatrix.get(m["n"], scale=False, dtype=torch.float64, device="cuda")
    got = hardmard_transform(to_device(G[f"{tag}|w"], m["dtype"]).double(), rhs, lhs, k).cpu().numpy()
    want = G[f"{tag}|hadamard_unscaled_f64"]
    assert np.abs(got - want).max() <= 1e-14 * max(1.0, np.abs(want).max())


def test_norm_folding_fixture():
    ln = nn.LayerNorm(64, bias=True, device="cuda")
    ln.weight.data = torch.from_numpy(G["norm|ln_w"]).cuda()
    ln.bias.data = torch.from_numpy(G["norm|ln_b"]).cuda()
    fc = nn.Linear(64, 32, bias=False, device="cuda", dtype=torch.bfloat16)
    fc.weight.data = to_device(G["norm|fc_w"], "bf16")
    transform_rms_norm_and_linear(ln, [fc])
    assert ln.bias is None and torch.equal(ln.weight.data, torch.ones_like(ln.weight.data))
    assert_bit_equal(to_np(fc.weight.data), G["norm|fc_w_out"], "folded weight")
    assert_bit_equal(to_np(fc.bias.data), G["norm|fc_b_out"], "folded bias")


@pytest.mark.parametrize("rows,n,dt", [(300, 1024, "bf16"), (96, 1792, "f16"), (64, 14336, "bf16")])
def test_rotations_vs_oracle_larger(rows, n, dt):
    """Llama-sized channel counts (14336 = 28 x 512 uses the order-28 Paley matrix)."""
    g = torch.Generator().manual_seed(rows + n)
    w = qo.round_to(dt, (torch.randn(rows, n, generator=g) * 0.03).numpy())
    rhs64, lhs64, k = HadamardMatrix.get(n, scale=True, dtype=torch.float64)
    lin = nn.Linear(n, rows, bias=False, device="cuda", dtype=TORCH[dt])
    lin.weight.data = to_device(w, dt)
    hadamard_in_channels([lin])
    want = ro.cast_from_f64(ro.hadamard_transform(w.astype(np.float64), rhs64.numpy(), lhs64.numpy() if k > 1 else None,
                                                  k, True), dt)
    got = to_np(lin.weight.data)
    assert (got != want).mean() <= 1e-5, (got != want).mean()
    assert np.abs(got - want).max() <= np.abs(want).max() * 2.0 ** -7
    if n <= 2048:
        torch.manual_seed(n)
        rot = get_rotation_matrix(n, random=True)
        p = nn.Parameter(to_device(w, dt), requires_grad=False)
        rotate_in_channels(p, rotation=rot)
        want = ro.rotate_in_channels(w, dt, rot.numpy())
        got = to_np(p.data)
        assert (got != want).mean() <= 1e-5, (got != want).mean()
        # rotating back with the transpose recovers the weight up to the two roundings
        rotate_in_channels(p, rotation=rot.T.contiguous())
        assert np.abs(to_np(p.data) - w).max() <= np.abs(w).max() * 2.0 ** -6
