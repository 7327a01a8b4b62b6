"""Time one whole GPTQ sweep (all column blocks with their lazy updates) of a 4096 x 14336 weight -- down_proj of a
Llama-3-8B layer -- alone on the chip: python tools/quick_sweep_full.py   (under rocprofv3 --kernel-trace --stats for
the kernels behind it)"""
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
R, Cn, lazy = 4096, 14336, 4
W0 = torch.randn(R, Cn, device=dev) * 0.02
U = torch.triu(torch.randn(Cn, Cn, device=dev) * 0.001) + torch.eye(Cn, device=dev)
ncg = Cn // 128
s2 = torch.full((R, ncg), 0.005, device=dev)
z2 = torch.full((R, ncg), 8.0, device=dev)
perm = torch.arange(Cn, device=dev)
Q = torch.zeros_like(W0)
err = torch.empty(R, 128 * lazy, device=dev)
nbytes = lib().dcq_gptq_sweep_workspace_bytes(R, Cn, lazy)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
W = W0.clone()


def run():
    W.copy_(W0)
    check(lib().dcq_gptq_sweep(C.byref(step), ptr(W), Cn, R, Cn, ptr(U), ptr(s2), ptr(z2), 0.0, 1, 128, ncg, ptr(perm), ptr(Q),
                               ptr(err), lazy, ptr(ws), nbytes, stream_of(dev)), "sweep")


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"sweep of {R} x {Cn}, lazy groups of {lazy} blocks: {e0.elapsed_time(e1) / reps:.3f} ms (with a 235 MB copy of the weight)")
