"""Kernel-only timing of the fused fake-quant (outputs preallocated, direct C-ABI calls)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcompressor_amd import engine, _lib
from deepcompressor_amd._lib import lib, ptr, storage_of, stream_of
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

def run(R, Cn, dt, name, zd, g, iters=50):
    w = (torch.randn(R, Cn, device="cuda") * 0.02).to(dt)
    q = Quantizer(QuantizerConfig(dtype=getattr(QDType, name), zero_point=zd, group_shapes=((1, g),), scale_dtypes=(torch.float16,)))
    info = q.update(w.shape, default_dtype=dt)
    step = info.steps[0].scale.native(w.device)
    x2d, _ = engine.canonical_groups(w, info.steps[0].tensor_view_shape)
    G, S = x2d.shape
    dq = torch.empty_like(x2d); sc = torch.empty(G, device="cuda"); ze = torch.empty(G, device="cuda")
    st = stream_of(w.device)
    f = lambda: lib().dcq_fake_quant_groups(C.byref(step), ptr(x2d), storage_of(dt), G, S, None, None, 0, 0.0, ptr(dq), None, ptr(sc), ptr(ze), None, st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = R * Cn * w.element_size() * 2 + G * 8
    print(f"{R}x{Cn} {str(dt)[6:]} {name} g{g}: {ms*1e3:8.1f} us  {nbytes/ms/1e6:7.0f} GB/s  ({nbytes/ms/1e6/8000*100:.1f}% of 8 TB/s)")

for (R, Cn) in ((4096, 4096), (14336, 4096), (32768, 8192)):
    for dt in (torch.bfloat16, torch.float16):
        run(R, Cn, dt, "sint4", None, 128)
        run(R, Cn, dt, "uint4", ZeroPointDomain.PostScale, 128)
run(32768, 8192, torch.bfloat16, "sint8", None, -1)
run(32768, 8192, torch.bfloat16, "sint4", None, 64)
run(16384, 8192, torch.float32, "sint4", None, 128)
