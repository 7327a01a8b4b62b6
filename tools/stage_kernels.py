"""The HBM / VALU-bound stage kernels on their own, as a plain program for rocprofv3 passes:
    fused fake-quantization of a 32768 x 8192 bf16 tensor (1.07 GB: past the 256 MiB Infinity Cache), 5 launches
    the NVFP4 (max reduction + value-table kernel) and QoQ W4 (both progressive steps in one kernel) passes on the same
    tensor, 5 launches each
    101-candidate range search on 14336 x 4096 bf16 (Llama-3-8B gate_proj), 3 launches
python3 tools/stage_kernels.py            (GPU box; rocprofv3 ... -- python3 tools/stage_kernels.py)
"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.torch = torch  # (bench imports torch inside main(), which this program does not run)

from deepcompressor_amd.calib import calibrate_dynamic_range  # noqa: E402
from deepcompressor_amd.data.common import TensorType  # noqa: E402

from deepcompressor_amd import engine  # noqa: E402

engine.FINITE_CHECKS = "deferred"  # launches back to back; the flag is looked at once at the end
dev = torch.device("cuda", 0)
Quantizer, qcfg, rcfg, _ = bench.build_objects()
q = Quantizer(qcfg, key="rate", develop_dtype=torch.float32)
big = (torch.randn(32768, 8192, device=dev) * 0.02).to(torch.bfloat16)
w = (torch.randn(14336, 4096, device=dev) * 0.02).to(torch.bfloat16)
for _ in range(5):
    q.quantize(big)
torch.cuda.synchronize()
from deepcompressor_amd.data.dtype import QuantDataType  # noqa: E402
from deepcompressor_amd.data.zero import ZeroPointDomain  # noqa: E402
from deepcompressor_amd.quantizer import ProgressiveQuantizerConfig, QuantizerConfig  # noqa: E402

nvfp4 = Quantizer(QuantizerConfig(dtype=QuantDataType.from_str("sfp4_e2m1_all"), zero_point=None,
                                  group_shapes=((-1, -1), (1, 16)),
                                  scale_dtypes=(None, QuantDataType.from_str("sfp8_e4m3_nan"))), key="nvfp4")
qoq = Quantizer(ProgressiveQuantizerConfig(
    dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale, group_shapes=((1, -1), (1, 128)),
    scale_dtypes=(torch.float16, QuantDataType.from_str("sint8")),
    intermediate_dtypes=(QuantDataType.from_str("sint8"),), intermediate_levels=(0,)), key="qoq")
for quantizer in (nvfp4, qoq):
    for _ in range(5):
        quantizer.quantize(big)
    torch.cuda.synchronize()
del big
for _ in range(3):
    calibrate_dynamic_range(TensorType.Weights, rcfg, True, q, modules=[None], activations=None,
                            weights=[nn.Parameter(w, requires_grad=False)])
torch.cuda.synchronize()
engine.raise_if_nonfinite()
print("done")
