"""Streaming rate of the fused fake-quant pass (HIP events around back-to-back calls) for a few configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

dev = torch.device("cuda:0")
cfgs = {"uint4_post": QuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 128),),
                                      scale_dtypes=(torch.float16,))}
for shape in ((4096, 4096), (14336, 4096), (32768, 8192)):
    w = (torch.randn(*shape, device=dev) * 0.02).to(torch.bfloat16)
    for name, cfg in cfgs.items():
        q = Quantizer(cfg, develop_dtype=torch.float32)
        for _ in range(3):
            q.quantize(w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            q.quantize(w)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        b = w.numel() * 4.0625
        print(f"nt={os.environ.get('DCQ_FUSED_NT', '0')} cap={os.environ.get('DCQ_FUSED_GRID_CAP', '-')} {shape} {name}: "
              f"{t * 1e6:8.1f} us  {b / t / 1e12:.2f} TB/s", flush=True)
