"""``dcq_cholesky_inverse_upper_static``: the inverse-factor routine for callers that cannot allocate (its workspace is
an array built into the library).  Called through the C ABI next to ``dcq_cholesky_inverse_upper`` -- same kernels, so
the bar is bit equality -- on two calls in a row, the second with a smaller matrix (the built-in array keeps the first
call's contents), and rejected beyond its size limit."""

import pytest
import torch

from deepcompressor_amd._lib import lib, ptr, stream_of

pytestmark = pytest.mark.gpu


def spd(n: int, seed: int, dev) -> torch.Tensor:
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.randn((n, n + 64), generator=g, device=dev)
    h = (a @ a.T) / (n + 64)
    h[5, :] = 0
    h[:, 5] = 0  # a dead channel: its zero diagonal entry becomes 1
    return h.contiguous()


def both(h: torch.Tensor, perm, diag_add: float):
    n, dev = h.shape[0], h.device
    out = []
    nbytes = lib().dcq_cholesky_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for static in (False, True):
        u = torch.full((n, n), float("nan"), device=dev)
        info = torch.full((1,), -1, dtype=torch.int32, device=dev)
        if static:
            rc = lib().dcq_cholesky_inverse_upper_static(ptr(h), n, ptr(perm), diag_add, ptr(u), ptr(info), stream_of(dev))
        else:
            rc = lib().dcq_cholesky_inverse_upper(ptr(h), n, ptr(perm), diag_add, ptr(u), ptr(ws), ws.numel(), ptr(info),
                                                  stream_of(dev))
        assert rc == 0, lib().dcq_last_error()
        torch.cuda.synchronize()
        out.append((u, int(info.item())))
    return out


def test_static_scratch_entry_point_equals_the_workspace_one():
    dev = torch.device("cuda", 0)
    for n, seed, use_perm in ((2048, 1, True), (1000, 2, False), (333, 3, True)):  # big first, then smaller ones
        h = spd(n, seed, dev)
        perm = torch.randperm(n, device=dev) if use_perm else None
        (u_ws, info_ws), (u_st, info_st) = both(h, perm, 0.01)
        assert info_ws == 0 and info_st == 0
        assert torch.isfinite(u_st).all()
        assert torch.equal(u_ws, u_st), f"n={n}: static-scratch result differs from the workspace entry point"
        # and it is the factor it claims to be: U^T U = S^-1
        hp = h.clone()
        hp[5, 5] = 1.0
        if perm is not None:
            hp = hp[perm][:, perm]
        s = hp.double() + 0.01 * torch.eye(n, dtype=torch.float64, device=dev)
        resid = (u_st.double().T @ u_st.double()) @ s - torch.eye(n, dtype=torch.float64, device=dev)
        assert resid.abs().max() < 5e-3, resid.abs().max().item()
    # a failing factorisation reports the same pivot through both entry points
    bad = spd(512, 9, dev)
    bad[100, 100] = -1.0
    (_, info_ws), (_, info_st) = both(bad, None, 0.0)
    assert info_ws > 0 and info_ws == info_st


def test_static_scratch_entry_point_rejects_sizes_beyond_its_array():
    dev = torch.device("cuda", 0)
    h = torch.eye(2049, device=dev)
    u = torch.empty_like(h)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib().dcq_cholesky_inverse_upper_static(ptr(h), 2049, None, 0.0, ptr(u), ptr(info), stream_of(dev))
    assert rc == -2  # DCQ_ERR_UNSUPPORTED
    assert b"2048" in lib().dcq_last_error()
