"""Time of the low-rank branch's leading singular triplets on a 3072 x 3072 bf16 weight (rank 32), host wall clock per
call; run under `rocprofv3 --kernel-trace --stats` for the kernels behind it."""
import sys, time
sys.path.insert(0, ".")
import torch
from deepcompressor_amd.nn.patch.lowrank import leading_singular_triplets

torch.manual_seed(0)
w = (torch.randn(3072, 3072, device="cuda") * 0.02 + torch.randn(3072, 48, device="cuda") @ torch.randn(48, 3072, device="cuda") * 0.01).to(torch.bfloat16)
for _ in range(2):
    leading_singular_triplets(w, 32)
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(reps):
    leading_singular_triplets(w, 32)
torch.cuda.synchronize()
print(f"leading_singular_triplets 3072^2 rank 32: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call")
