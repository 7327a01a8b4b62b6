"""Timings that size the top-r eigen-solve: small eigh / cholesky in torch, tall-skinny float64 products."""
import sys, time
sys.path.insert(0, ".")
import torch
from deepcompressor_amd.utils.math.hadamard import gemm_f64

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

dev = "cuda"
for k in (64, 128, 256, 512):
    a = torch.randn(k, k, dtype=torch.float64, device=dev); a = a @ a.T + k * torch.eye(k, dtype=torch.float64, device=dev)
    print(f"k={k}: eigh {t(lambda: torch.linalg.eigh(a)):.2f} ms  cholesky {t(lambda: torch.linalg.cholesky(a)):.2f} ms  "
          f"qr {t(lambda: torch.linalg.qr(torch.randn(3072, k, dtype=torch.float64, device=dev))):.2f} ms", flush=True)
n = 3072
g = torch.randn(n, n, dtype=torch.float64, device=dev); g = (g + g.T).contiguous()
for k in (64, 128, 256, 512):
    q = torch.randn(n, k, dtype=torch.float64, device=dev)
    ms = t(lambda: gemm_f64(g, q, torch.float64), 10)
    ms2 = t(lambda: g @ q, 10)
    ms3 = t(lambda: gemm_f64(q, q, torch.float64, a_transposed=True), 10)
    print(f"G({n}^2) @ Q({n}x{k}): dcq {ms:.3f} ms ({2*n*n*k/ms/1e9:.1f} TFLOP/s), torch {ms2:.3f} ms; Q^T Q dcq {ms3:.3f} ms", flush=True)
w = torch.randn(n, n, device=dev).to(torch.bfloat16)
print(f"gram W^T W: {t(lambda: gemm_f64(w, w, torch.float64, a_transposed=True), 5):.3f} ms; eigh(3072) {t(lambda: torch.linalg.eigh(g), 2):.1f} ms")
