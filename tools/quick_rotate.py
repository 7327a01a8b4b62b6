"""Time the float64 rotation GEMM: python tools/quick_rotate.py"""
import sys
sys.path.insert(0, ".")
import torch
from deepcompressor_amd.utils.math.hadamard import gemm_f64

for (M, N, K, dt) in [(4096, 4096, 4096, torch.bfloat16), (14336, 4096, 4096, torch.bfloat16), (4096, 14336, 14336, torch.bfloat16),
                      (4096 * 28, 512, 512, torch.bfloat16)]:
    a = torch.randn(M, K, device="cuda").to(dt)
    b = torch.randn(K, N, device="cuda", dtype=torch.float64)
    gemm_f64(a, b, dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gemm_f64(a, b, dt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"rotate M={M} N={N} K={K}: {ms:.2f} ms  {2 * M * N * K / ms / 1e9:.1f} TF/s (f64 MFMA peak 78.6)", flush=True)
    if M * N <= 4096 * 4096:
        ref = (a.double() @ b).to(dt)
        torch.cuda.synchronize()
        e0.record(); ref = (a.double() @ b).to(dt); e1.record(); torch.cuda.synchronize()
        print(f"   torch (widen + matmul + cast): {e0.elapsed_time(e1):.2f} ms; mismatching elements {(ref != gemm_f64(a, b, dt)).float().mean().item():.2e}")
