"""Checksums and timings of the inverse factor for a few sizes (compare two builds of the library: DCQ_LIB=...)."""
import hashlib
import os
import sys
sys.path.insert(0, ".")
import torch
if os.environ.get("DCQ_LIB"):
    import deepcompressor_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["DCQ_LIB"])
from deepcompressor_amd.quantizer.kernel.gptq import cholesky_inverse_upper

for n in (128, 200, 256, 640, 2048, 4096):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2 * n, n, generator=g) * torch.exp(0.5 * torch.randn(n, generator=g))
    h = ((x.T @ x) / n).cuda()
    perm = torch.randperm(n, generator=g).cuda() if n != 256 else None
    u, tries = cholesky_inverse_upper(h, perm, float(h.diagonal().mean()), 0.01, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        cholesky_inverse_upper(h, perm, float(h.diagonal().mean()), 0.01, 3)
    e1.record()
    torch.cuda.synchronize()
    print(n, tries, hashlib.sha256(u.cpu().numpy().tobytes()).hexdigest()[:16], f"{e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
