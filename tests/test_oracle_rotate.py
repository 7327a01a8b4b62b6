"""Rotation oracle and the host-side Hadamard constructions against fixtures produced by the reference
(oracle/make_golden.py rotate).  CPU only."""

import numpy as np
import pytest
import torch
from conftest import assert_bit_equal, load_golden

import rotate_oracle as ro
from deepcompressor_amd.utils.math.hadamard import HadamardMatrix, _hadamard_k, _matmul_hadU, random_hadamard_matrix

G = load_golden("rotate")
CASES = [k for k, v in G.meta.items() if "n" in v]
OUT_CASES = [k for k, v in G.meta.items() if "heads" in v]


@pytest.mark.parametrize("k", [12, 20, 28, 36, 40, 52, 60, 108, 140, 156, 172])
def test_constructed_hadamard_equals_reference_table(k):
    """Paley / Williamson / doubling constructions reproduce the reference's literal tables entry by entry."""
    want = np.unpackbits(G[f"hadamard|{k}"])[: k * k].reshape(k, k).astype(np.int8) * 2 - 1
    got = _hadamard_k(k)
    assert np.array_equal(got, want)
    assert np.array_equal(got.astype(np.int64) @ got.T.astype(np.int64), k * np.eye(k, dtype=np.int64))
    assert torch.equal(HadamardMatrix._get_hadamard_k(k), torch.from_numpy(want.astype(np.float32)))


@pytest.mark.parametrize("tag", CASES)
def test_rotation_matrices_match_reference(tag):
    from deepcompressor_amd.calib.rotate import get_rotation_matrix

    m = G.meta[tag]
    fixed = get_rotation_matrix(m["n"], random=False)
    assert np.array_equal(fixed.numpy(), G[f"{tag}|rotation_fixed"])
    torch.manual_seed(m["seed_random"])
    rand = get_rotation_matrix(m["n"], random=True)
    assert np.array_equal(rand.numpy(), G[f"{tag}|rotation_random"])
    # orthogonality (float64)
    r = rand.numpy()
    assert np.abs(r @ r.T - np.eye(m["n"])).max() < 1e-6  # the reference scales by a float32 square root


@pytest.mark.parametrize("tag", CASES)
def test_oracle_rotate_in_and_hadamard(tag):
    m = G.meta[tag]
    w = G[f"{tag}|w"]
    for name in ("fixed", "random"):
        got = ro.rotate_in_channels(w, m["dtype"], G[f"{tag}|rotation_{name}"])
        assert_bit_equal(got, G[f"{tag}|in_{name}"], f"{tag} in_{name}")
    k = m["k"]
    lhs = _hadamard_k(k).astype(np.float64) if k > 1 else None
    rhs = ro.sylvester(m["n"] // k)
    scaled = ro.hadamard_transform(w.astype(np.float64), rhs * (1.0 / np.sqrt(np.float64(m["n"]))), lhs, k, True)
    assert_bit_equal(ro.cast_from_f64(scaled, m["dtype"]), G[f"{tag}|hadamard_in"], f"{tag} hadamard_in")
    unscaled = ro.hadamard_transform(w.astype(np.float64), rhs, lhs, k, False)
    assert np.abs(unscaled - G[f"{tag}|hadamard_unscaled_f64"]).max() <= 1e-15 * max(1.0, np.abs(unscaled).max()) * 8


@pytest.mark.parametrize("tag", OUT_CASES)
def test_oracle_rotate_out(tag):
    m = G.meta[tag]
    rot = G[f"{tag}|rotation"]
    assert_bit_equal(ro.rotate_out_channels(G[f"{tag}|w"], m["dtype"], rot), G[f"{tag}|w_out"], f"{tag} w")
    got_b = ro.rotate_out_channels(G[f"{tag}|b"].reshape(-1, 1), m["dtype"], rot).reshape(-1)
    assert_bit_equal(got_b, G[f"{tag}|b_out"], f"{tag} b")


def test_butterflies_restate_matmul_hadU():
    g = torch.Generator().manual_seed(0)
    for n in (64, 96, 224):
        x = torch.randn(7, n, generator=g, dtype=torch.float64)
        hk, k = HadamardMatrix.get_lhs(n)
        want = _matmul_hadU(x, hk, k).numpy()
        got = ro.butterflies(x.numpy(), hk.numpy() if k > 1 else None, k)
        assert np.array_equal(got, want)
    torch.manual_seed(5)
    q = random_hadamard_matrix(96).numpy()
    assert np.abs(q @ q.T - np.eye(96)).max() < 1e-6  # float32 square root in the scaling: orthogonal to ~1e-8
