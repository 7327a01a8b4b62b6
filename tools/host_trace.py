"""Host-side timeline of one bench step: when each host thread enters and leaves the stage's main calls, relative to the
step's start (is a pipeline late because its host thread is?): python tools/host_trace.py [steps]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.torch = torch
from deepcompressor_amd.app import llm  # noqa: E402
from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod  # noqa: E402

log, origin = [], [0.0]


def traced(owner, name):
    fn = getattr(owner, name)

    def wrapper(*args, **kwargs):
        t0 = time.perf_counter()
        try:
            return fn(*args, **kwargs)
        finally:
            log.append((threading.current_thread().name, name, (t0 - origin[0]) * 1e3, (time.perf_counter() - origin[0]) * 1e3))

    setattr(owner, name, wrapper)


for owner, name in [(gptq_mod, "_inverse_factor_of"), (gptq_mod, "_enqueue_factor"), (gptq_mod, "gptq_quantize"),
                    (gptq_mod, "_workspace"), (gptq_mod, "prepare_hessians"), (gptq_mod, "_diag_mean"),
                    (llm.LlmWeightQuantizer, "calibrate_dynamic_range"), (llm, "_quantize_layer_weights")]:
    traced(owner, name)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
bench.build_objects()
config = bench.stage_config()
weights, caches = bench.make_layer(device, 262144, seed=1234)
for i in range(n):
    torch.cuda.synchronize()
    log.clear()
    origin[0] = time.perf_counter()
    bench.quantize_layer(weights, caches, config, streams=4)
    enqueued = (time.perf_counter() - origin[0]) * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - origin[0]) * 1e3
print(f"last step: enqueued after {enqueued:.1f} ms, done after {total:.1f} ms")
for thread, name, t0, t1 in sorted(log, key=lambda r: r[2]):
    if t1 - t0 >= 0.2 or name in ("_enqueue_factor", "_inverse_factor_of"):
        print(f"{thread:14s} {t0:8.2f} -> {t1:8.2f}  {t1 - t0:7.2f} ms  {name}")
