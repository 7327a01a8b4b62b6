"""Experiment: two layers in flight (each from its own host thread and main stream, the second started half a step
later) against the same layers one after the other.  Upper bound of what overlapping a layer's tail with the next
layer's Hessians could give: python tools/exp_two_layers.py [steps] [delay_ms]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.torch = torch
from deepcompressor_amd import parallel
from deepcompressor_amd.quantizer.kernel import gptq as gptq_mod

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
bench.build_objects()
config = bench.stage_config()
sets = [bench.make_layer(device, 128 * 2048, seed=1234 + i) for i in range(2)]
tls = threading.local()

orig_job_stream = parallel._job_stream
def job_stream(dev, slot, prio):
    return orig_job_stream(dev, slot + 10 * getattr(tls, "index", 0), prio)
parallel._job_stream = job_stream

def clear_own(keep_ahead=False):
    gptq_mod._state.hessian["key"] = gptq_mod._state.hessian["refs"] = gptq_mod._state.hessian["value"] = None
    gptq_mod._state.factor["key"] = gptq_mod._state.factor["value"] = None
    own = {id(t) for t in tls.tensors}
    with gptq_mod._prepared_lock:
        for key in [k for k, (refs, *_) in gptq_mod._prepared.items() if any(id(t) in own for t in refs)]:
            del gptq_mod._prepared[key]
gptq_mod.clear_hessian_cache = clear_own

def run(index, n, start_delay=0.0):
    tls.index = index
    weights, caches = sets[index]
    tls.tensors = [t for c in caches.values() for t in c.data]
    torch.cuda.set_device(device)
    stream = torch.cuda.Stream(device)
    time.sleep(start_delay)
    with torch.inference_mode(), torch.cuda.stream(stream):
        for _ in range(n):
            bench.quantize_layer(weights, caches, config, streams=4)
    stream.synchronize()

for i in range(2):
    run(i, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2):
    run(i, steps)
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / (2 * steps) * 1e3
t0 = time.perf_counter()
threads = [threading.Thread(target=run, args=(i, steps, i * delay * 1e-3)) for i in range(2)]
for t in threads: t.start()
for t in threads: t.join()
torch.cuda.synchronize()
par = (time.perf_counter() - t0) / (2 * steps) * 1e3
print(f"one layer at a time {seq:.1f} ms per layer; two in flight (second {delay:.0f} ms later) {par:.1f} ms per layer")
