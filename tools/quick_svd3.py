import sys, time
sys.path.insert(0, ".")
import torch
from deepcompressor_amd.nn.patch import lowrank as lr
g = torch.Generator().manual_seed(1)
w = (torch.randn(3072, 3072, generator=g) * 0.02).to(torch.bfloat16).cuda()
lr.leading_singular_triplets(w, 32); torch.cuda.synchronize()
calls = {}
def wrap(name):
    f = getattr(lr, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); calls.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    setattr(lr, name, g)
if len(sys.argv) > 1:
    for n in ("_orthonormalize", "_chebyshev_filter"):
        wrap(n)
    eigh = torch.linalg.eigh
    def timed_eigh(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = eigh(*a, **k); torch.cuda.synchronize()
        calls.setdefault("eigh", []).append(time.perf_counter() - t0); return r
    torch.linalg.eigh = timed_eigh
t0 = time.perf_counter()
lr.leading_singular_triplets(w, 32); torch.cuda.synchronize()
print(f"total {(time.perf_counter() - t0) * 1e3:.1f} ms")
for k, v in calls.items():
    print(k, len(v), f"{sum(v) * 1e3:.2f} ms", [round(x * 1e3, 2) for x in v])
