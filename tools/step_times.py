"""Per-step wall time of the bench's layer step from a cold process (does the step settle after the default two
warm-up steps?): python tools/step_times.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.torch = torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
bench.build_objects()
config = bench.stage_config()
weights, caches = bench.make_layer(device, 262144, seed=1234)
times = []
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.quantize_layer(weights, caches, config, streams=4)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", " ".join(f"{t:.1f}" for t in times))
