"""Streaming rate of the single-pass fake-quantization variants on a 1.07 GB bf16 tensor (32768 x 8192), with the
NaN / Inf look deferred to the end of each timed loop (the calls then do not block):
python tools/fq_variants_rate.py [case ...]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch
from _product import build_quantizer
from conftest import load_golden

from deepcompressor_amd import engine

engine.FINITE_CHECKS = "deferred"
FQ = load_golden("fake_quant")
x = (torch.randn(32768, 8192, device="cuda") * 0.02).to(torch.bfloat16)
n = x.numel()
for name in (sys.argv[1:] or ("uint4_post_g128_f16s", "nvfp4_g16", "qoq_g128")):
    q = build_quantizer(FQ.meta["cases"][name])
    q.quantize(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for _ in range(reps):
        q.quantize(x)
    e1.record()
    torch.cuda.synchronize()
    engine.raise_if_nonfinite()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f"{name}: {t * 1e6:.0f} us per call, {n * 4 / t / 1e12:.2f} TB/s of the 4 B/param a single pass moves", flush=True)
