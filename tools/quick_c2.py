"""QoQ W4 progressive fake-quant of the 7 linears of a Llama-3-8B layer: wall per layer under both NaN / Inf policies,
the kernels' own time (events around the loop), and the host time per call."""
import sys
import time

import torch

sys.path.insert(0, ".")
from deepcompressor_amd import engine  # noqa: E402
from deepcompressor_amd.data.dtype import QuantDataType  # noqa: E402
from deepcompressor_amd.data.zero import ZeroPointDomain  # noqa: E402
from deepcompressor_amd.quantizer import Quantizer  # noqa: E402
from deepcompressor_amd.quantizer.config.base import ProgressiveQuantizerConfig  # noqa: E402

SHAPES = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)]
qoq = Quantizer(ProgressiveQuantizerConfig(
    dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale, group_shapes=((1, -1), (1, 128)),
    scale_dtypes=(torch.float16, QuantDataType.from_str("sint8")),
    intermediate_dtypes=(QuantDataType.from_str("sint8"),), intermediate_levels=(0,)), key="c2")
layer = [(torch.randn(r, c, device="cuda") * 0.02).to(torch.float16) for r, c in SHAPES]


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for policy in ("always", "deferred"):
    engine.FINITE_CHECKS = policy
    tail = (lambda: engine.raise_if_nonfinite()) if policy == "deferred" else (lambda: None)
    print(policy, "layer ms", wall(lambda: ([qoq.quantize(x) for x in layer], tail()), 50))
engine.FINITE_CHECKS = "deferred"
print("deferred, one look per 50 layers: layer ms", wall(lambda: [qoq.quantize(x) for x in layer], 50))
t0 = time.perf_counter()
for _ in range(20):
    outs = [qoq.quantize(x) for x in layer]
host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print("host time to issue one layer (GPU behind) ms", host * 1e3)
del outs
for x in layer:
    t = wall(lambda: qoq.quantize(x), 50)
    print(tuple(x.shape), "us per call", t * 1e3, "TB/s", x.numel() * 4 / t / 1e9)
small = layer[1][:64].contiguous()
print("host us per call (64 x 4096)", wall(lambda: qoq.quantize(small), 500) * 1e3)
engine.FINITE_CHECKS = "always"
print("host us per call, always (64 x 4096)", wall(lambda: qoq.quantize(small), 500) * 1e3)
