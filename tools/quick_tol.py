"""Leading singular triplets of a 3072 x 3072 weight at several residual bars of the block iteration: time, rounds of
Rayleigh-Ritz, and the error of b @ a and of the singular values against the full float64 SVD."""
import sys
import time

import torch

sys.path.insert(0, ".")
from deepcompressor_amd.nn.patch import lowrank as lr  # noqa: E402

g = torch.Generator().manual_seed(1)
cases = {"gaussian": torch.randn(3072, 3072, generator=g) * 0.02}
u, _ = torch.linalg.qr(torch.randn(3072, 64, generator=g))
v, _ = torch.linalg.qr(torch.randn(3072, 64, generator=g))
cases["planted"] = cases["gaussian"] + (u * torch.logspace(1, -1, 64)) @ v.t()
inner = lr._leading_eigenpairs
for name, w32 in cases.items():
    w = w32.to(torch.bfloat16).cuda()
    U, S, Vh = torch.linalg.svd(w.double())
    ref = (U[:, :32] * S[:32]) @ Vh[:32]
    for passes in (2, 3, 4):
      lr.FILTER_PASSES = passes
      for tol in (1e-9,):
        rounds = []
        eigh = torch.linalg.eigh

        def counted(*a, **k):
            rounds.append(1)
            return eigh(*a, **k)

        torch.linalg.eigh = counted
        lr._leading_eigenpairs = lambda gram, rank, **kw: inner(gram, rank, tol=tol)
        lr.leading_singular_triplets(w, 32)
        torch.cuda.synchronize()
        rounds.clear()
        t0 = time.perf_counter()
        us, vh = lr.leading_singular_triplets(w, 32)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        torch.linalg.eigh = eigh
        err = ((us @ vh) - ref).abs().max().item() / w.double().abs().max().item()
        sv = (us.norm(dim=0) / S[:32] - 1).abs().max().item()
        print(f"{name} passes {passes} tol {tol:g}: {ms:.1f} ms, {len(rounds)} small eigh, |b a - ref| / max|w| = {err:.2e}, sigma rel err {sv:.2e}", flush=True)
