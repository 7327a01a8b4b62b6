"""GPU timing of the float32 GEMM and the inverse-factor routine: python tools/quick_linalg.py"""
import sys
import torch
sys.path.insert(0, ".")
import os
if os.environ.get("DCQ_LIB"):  # timing experiments: another build of the library
    import deepcompressor_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["DCQ_LIB"])
from deepcompressor_amd.quantizer.kernel.gptq import cholesky_inverse_upper, gemm_f32


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if len(sys.argv) > 1 and sys.argv[1] == "err":
    # accuracy of the large-problem GEMM path against float64, beside torch's float32 GEMM
    torch.manual_seed(0)
    for (M, N, K, at) in [(3072, 3072, 1024, False), (3072, 3072, 1024, True), (3000, 3100, 512, False), (3001, 3103, 512, True),
                          (4096, 13824, 512, False)]:
        a = torch.randn((K, M) if at else (M, K), device="cuda") * torch.logspace(-3, 3, K, device="cuda").reshape((K, 1) if at else (1, K))
        b = torch.randn(K, N, device="cuda")
        c = torch.randn(M, N, device="cuda")
        am = a.T if at else a
        ref = c.double() - am.double() @ b.double()
        scale = (am.double().abs() @ b.double().abs())
        out = c.clone()
        gemm_f32(a, b, out, alpha=-1.0, beta=1.0, a_transposed=at)
        t = c.clone().addmm_(am, b, alpha=-1.0)
        e_ours = ((out.double() - ref).abs() / scale).max().item()
        e_torch = ((t.double() - ref).abs() / scale).max().item()
        e_two = float("nan")
        if M % 4 == 0 and N % 4 == 0:
            out2 = c.clone()
            gemm_f32(a, b, out2, alpha=-1.0, beta=1.0, a_transposed=at, two_term=True)
            e_two = ((out2.double() - ref).abs() / scale).max().item()
        print(f"M={M} N={N} K={K} at={at}: max |err| / (|A||B|)  ours {e_ours:.3e}  two-term {e_two:.3e}  torch f32 {e_torch:.3e}", flush=True)
    # rows of A and columns of B spread over 12 decades: the per-row / per-column scales must keep every entry's accuracy
    for (M, N, K, at) in [(1024, 1536, 512, False), (1024, 1536, 512, True)]:
        a = torch.randn((K, M) if at else (M, K), device="cuda") * torch.logspace(-6, 6, M, device="cuda").reshape((1, M) if at else (M, 1))
        b = torch.randn(K, N, device="cuda") * torch.logspace(6, -6, N, device="cuda").reshape(1, N)
        am = a.T if at else a
        ref = am.double() @ b.double()
        scale = (am.double().abs() @ b.double().abs())
        out2 = torch.empty(M, N, device="cuda")
        gemm_f32(a, b, out2, a_transposed=at, two_term=True)
        t = am @ b
        print(f"graded M={M} N={N} K={K} at={at}: two-term {((out2.double() - ref).abs() / scale).max().item():.3e}  "
              f"torch f32 {((t.double() - ref).abs() / scale).max().item():.3e}", flush=True)
    sys.exit(0)
only_inv = len(sys.argv) > 1 and sys.argv[1] in ("inv", "big")
sizes = [int(v) for v in sys.argv[2:]] if only_inv else (128, 1024, 4096, 14336)
for (M, N, K, at) in [(7168, 7168, 7168, False), (7168, 7168, 7168, True), (4096, 13824, 512, False)] if sys.argv[1:2] == ["big"] else [] if only_inv else [(4096, 14208, 128, False), (14336, 3968, 128, False), (4096, 4096, 128, False), (4096, 13824, 512, False), (14336, 3584, 512, False),
                      (7168, 7168, 7168, False), (7168, 7168, 7168, True), (2048, 2048, 2048, True),
                      (1024, 1024, 1024, True), (512, 512, 512, False), (128, 14208, 128, True), (128, 128, 128, True)]:
    a = torch.randn((K, M) if at else (M, K), device="cuda")
    b = torch.randn(K, N, device="cuda")
    c = torch.randn(M, N, device="cuda")
    ms = timed(lambda: gemm_f32(a, b, c, alpha=-1.0, beta=1.0, a_transposed=at))
    ms_t = timed(lambda: c.addmm_(a.T if at else a, b, alpha=-1.0))
    ms_2 = timed(lambda: gemm_f32(a, b, c, alpha=-1.0, beta=1.0, a_transposed=at, two_term=True)) if K % 32 == 0 else float("nan")
    print(f"gemm M={M} N={N} K={K} at={at}: {ms*1e3:9.1f} us {2*M*N*K/ms/1e9:7.1f} TF/s | two-term {ms_2*1e3:9.1f} us "
          f"{2*M*N*K/ms_2/1e9:7.1f} TF/s | torch addmm {ms_t*1e3:9.1f} us {2*M*N*K/ms_t/1e9:7.1f} TF/s", flush=True)

for n in sizes:
    x = torch.randn(2 * n, n, device="cuda")
    h = (x.T @ x) / n
    del x
    mean = float(h.diagonal().mean())
    ms = timed(lambda: cholesky_inverse_upper(h, None, mean, 0.01, 3), reps=3)
    print(f"inverse factor n={n}: {ms:9.3f} ms  ({(2/3)*n**3/ms/1e9:6.1f} TF/s on 2/3 n^3)", flush=True)
