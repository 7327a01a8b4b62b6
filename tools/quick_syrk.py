"""Ad-hoc SYRK timing: python tools/quick_syrk.py [T] [C...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcompressor_amd.quantizer.kernel.gptq import hessian_accumulate
from deepcompressor_amd import _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
Cs = [int(a) for a in sys.argv[2:]] or [4096, 14336]
for C in Cs:
    x = torch.randn(T, C, device="cuda").to(torch.bfloat16)
    H = torch.zeros(C, C, device="cuda")
    hessian_accumulate(H, x, 1.0)
    torch.cuda.synchronize()
    ref = (x[:, :256].float().T @ x[:, :512].float())
    err = ((H[:256, :512] - ref).abs().max() / ref.abs().max()).item()
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hessian_accumulate(H, x, 1.0, start_event=e0, stop_event=e1)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = min(ms)
    nt = ((C + 255) // 256) * ((C + 255) // 256 + 1) // 2
    print(f"C={C} T={T}: syrk {t:.2f} ms  algorithmic {2*T*C*C/t/1e9:.0f} TF  executed {2*T*nt*256*256/t/1e9:.0f} TF  relerr {err:.2e}")
