"""Per-group stage timings of the layer pipeline with the Hessians prepared (not the contract bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import bench
from deepcompressor_amd.calib import calibrate_dynamic_range
from deepcompressor_amd.data.common import TensorType
from deepcompressor_amd.quantizer.kernel import gptq as gm

dev = torch.device("cuda:0")
tokens = int(os.environ.get("TOKENS", 32768))
weights, caches = bench.make_layer(dev, tokens, 1)
Quantizer, qcfg, rcfg, gcfg = bench.build_objects()

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

for rep in range(2):
    for name, r, c, key in bench.LLAMA3_8B_LAYER:
        w = weights[name]
        gm.prepare_hessians([caches[key]], dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0 = ev()
        q = Quantizer(qcfg, key=name, kernel=gcfg, develop_dtype=torch.float32)
        ranges = calibrate_dynamic_range(TensorType.Weights, rcfg, True, q, modules=[None], activations=None,
                                         weights=[nn.Parameter(w, requires_grad=False)])
        e1 = ev()
        h = gm.hessian_of(caches[key], dev)
        gm._inverse_factor_of(h, gcfg)
        e2 = ev()
        out = q.quantize(w, dynamic_range=ranges[0], inputs=caches[key]).data
        e3 = ev(); torch.cuda.synchronize(); t1 = time.perf_counter()
        if rep:
            print(f"{name:10s} {r}x{c}: search {e0.elapsed_time(e1):7.2f}  factor {e1.elapsed_time(e2):7.2f}  "
                  f"quantize(sweep) {e2.elapsed_time(e3):7.2f}  wall {1e3*(t1-t0):7.2f} ms")
    gm.clear_hessian_cache()
