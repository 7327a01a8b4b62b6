"""Where the time between back-to-back quantize() calls goes: per-call event pairs vs one pair around all calls."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ctypes as C
import torch
from _product import build_quantizer
from conftest import load_golden
from deepcompressor_amd import engine, _lib
from deepcompressor_amd._lib import lib, ptr, storage_of, stream_of

engine.FINITE_CHECKS = "deferred"
FQ = load_golden("fake_quant")
x = (torch.randn(32768, 8192, device="cuda") * 0.02).to(torch.bfloat16)
q = build_quantizer(FQ.meta["cases"]["uint4_post_g128_f16s"])
for _ in range(3):
    q.quantize(x)
torch.cuda.synchronize()
reps = 40
pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in pairs:
    a.record(); q.quantize(x); b.record()
torch.cuda.synchronize()
inner = sum(a.elapsed_time(b) for a, b in pairs) / reps
outer = pairs[0][0].elapsed_time(pairs[-1][1]) / reps
print(f"quantize(): per-call event pairs {inner * 1e3:.1f} us, first-to-last {outer * 1e3:.1f} us per call")
# the C ABI directly on preallocated buffers
q.update(x.shape, x.dtype)
step = q.info.steps[-1].scale.native(x.device)
G, S = x.numel() // 128, 128
dq = torch.empty_like(x); sc = torch.empty(G, device="cuda"); ze = torch.empty(G, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
def call():
    _lib.check(lib().dcq_fake_quant_groups(C.byref(step), ptr(x), storage_of(x.dtype), G, S, None, None, 0, 0.0, ptr(dq), None,
                                           ptr(sc), ptr(ze), ptr(flag), stream_of(x.device)), "fq")
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
print(f"C ABI on preallocated buffers: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call")
y = torch.empty_like(x)
e0.record()
for _ in range(reps):
    y.copy_(x)
e1.record()
torch.cuda.synchronize()
print(f"torch copy_: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call")
