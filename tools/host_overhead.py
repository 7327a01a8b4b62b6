"""Host time per quantize() call (tiny tensors, so the GPU never is the bound): python tools/host_overhead.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from _product import build_quantizer
from conftest import load_golden
from deepcompressor_amd import engine

FQ = load_golden("fake_quant")
x = (torch.randn(64, 1024, device="cuda") * 0.02).to(torch.bfloat16)
for policy in ("deferred", "always"):
    engine.FINITE_CHECKS = policy
    for name in ("uint4_post_g128_f16s", "nvfp4_g16", "qoq_g128"):
        q = build_quantizer(FQ.meta["cases"][name])
        for _ in range(20):
            q.quantize(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            q.quantize(x)
        torch.cuda.synchronize()
        print(f"{policy:9s} {name}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call", flush=True)
if len(sys.argv) > 1:
    import cProfile, pstats
    engine.FINITE_CHECKS = "deferred"
    q = build_quantizer(FQ.meta["cases"][sys.argv[1] if sys.argv[1] in FQ.meta["cases"] else "uint4_post_g128_f16s"])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        q.quantize(x)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
