"""leading_singular_triplets on FLUX-sized weights: time and accuracy against the float64 SVD."""
import sys, time
sys.path.insert(0, ".")
import torch
from deepcompressor_amd.nn.patch import lowrank as lr

def run(shape, kind):
    g = torch.Generator().manual_seed(shape[0] + shape[1])
    o, i = shape
    if kind == "planted":
        w = (torch.randn(o, 24, generator=g) @ torch.randn(24, i, generator=g) * 0.02 + torch.randn(o, i, generator=g) * 0.02)
    elif kind == "flat":
        w = torch.randn(o, i, generator=g) * 0.02
    else:  # heavy tailed
        w = torch.randn(o, i, generator=g) * 0.02 * torch.exp(torch.randn(i, generator=g))
    w = w.to(torch.bfloat16).cuda()
    lr.leading_singular_triplets(w, 32); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        us, vh = lr.leading_singular_triplets(w, 32)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    s_full = torch.linalg.svdvals(w.double())
    sigma = us.norm(dim=0)
    rel = ((sigma - s_full[:32]).abs() / s_full[:32]).max().item()
    resid = (w.double() - us @ vh).norm().item(); tail = s_full[32:].norm().item()
    orth = (vh @ vh.T - torch.eye(32, device="cuda", dtype=torch.float64)).abs().max().item()
    print(f"{shape} {kind}: {ms:.1f} ms  sigma rel {rel:.2e}  resid/tail-1 {resid/tail-1:.2e}  orth {orth:.1e}", flush=True)

for shape in ((3072, 3072), (1536, 4608), (4608, 1536), (12288, 3072)):
    for kind in ("planted", "flat", "heavy"):
        run(shape, kind)
