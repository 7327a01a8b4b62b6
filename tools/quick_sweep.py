"""Time the GPTQ sweep kernel alone for several row counts / block widths: python tools/quick_sweep.py"""
import ctypes as C
import sys
sys.path.insert(0, ".")
import torch
from deepcompressor_amd import engine
from deepcompressor_amd._lib import lib, check, ptr, stream_of
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.range import QuantRange
from deepcompressor_amd.data.zero import ZeroPointDomain

dev = torch.device("cuda")
dt = QDType.uint4
qr = QuantRange.construct(dt, has_zero_point=True, quant_range=None)
step = engine.make_step(quant_dtype=dt, zero_domain=ZeroPointDomain.PostScale, tensor_quant_range=qr, quant_range=None,
                        scale_dtype=torch.float32, zero_dtype=torch.float32, range_bound=None, q_span=1.0, device=dev)
for R, Cn in [(1024, 4096), (4096, 4096), (14336, 4096), (4096, 14336)]:
    W = torch.randn(R, Cn, device=dev) * 0.02
    U = torch.triu(torch.randn(Cn, Cn, device=dev) * 0.01) + torch.eye(Cn, device=dev)
    ncg = Cn // 128
    s2 = torch.full((R, ncg), 0.005, device=dev)
    z2 = torch.full((R, ncg), 0.04, device=dev)
    perm = torch.randperm(Cn, device=dev)
    Q = torch.zeros_like(W)
    err = torch.empty(R, 512, device=dev)
    for bc in (128, 64, 8):
        def run():
            check(lib().dcq_gptq_sweep_block(C.byref(step), ptr(W), Cn, R, Cn, ptr(U), 256, bc, ptr(s2), ptr(z2), 0.0, 1, 128,
                                             ncg, ptr(perm), ptr(Q), ptr(err), 512, stream_of(dev)), "sweep")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        print(f"R={R} C={Cn} bc={bc}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
