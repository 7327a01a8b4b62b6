"""Timing of the float64 SVD / eigendecomposition routes for the low-rank branch (not the contract bench)."""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
w = torch.randn(m, n, device="cuda", dtype=torch.float64)
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print(f"svd full {m}x{n} f64: {timed(lambda: torch.linalg.svd(w, full_matrices=False)):.3f} s", flush=True)
g = w.T @ w
print(f"eigh {n} f64: {timed(lambda: torch.linalg.eigh(g)):.3f} s", flush=True)
print(f"gram f64 matmul: {timed(lambda: w.T @ w)*1e3:.2f} ms", flush=True)
q = torch.randn(n, 96, device="cuda", dtype=torch.float64)
print(f"tall qr {n}x96 f64: {timed(lambda: torch.linalg.qr(q))*1e3:.2f} ms", flush=True)
print(f"W@Q {m}x{n}x96: {timed(lambda: w @ q)*1e3:.3f} ms", flush=True)
sys.path.insert(0, ".")
from deepcompressor_amd.nn.patch.lowrank import LowRankBranch
for shape in ((3072, 3072), (9216, 3072), (12288, 3072), (3072, 12288)):
    wb = (torch.randn(*shape, device="cuda") * 0.02).to(torch.bfloat16)
    t = timed(lambda: LowRankBranch(shape[1], shape[0], 32, weight=wb))
    print(f"LowRankBranch rank 32 on {shape[0]}x{shape[1]} bf16 (Gram route): {t*1e3:.1f} ms", flush=True)
