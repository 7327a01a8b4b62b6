"""Ad-hoc kernel timings (not the contract bench): python tools/quick_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcompressor_amd import engine
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig
from deepcompressor_amd.calib import DynamicRangeCalibConfig
from deepcompressor_amd.calib.config import SearchBasedCalibStrategy

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

R, C = int(os.environ.get("R", 4096)), int(os.environ.get("C", 4096))
for dt in (torch.bfloat16, torch.float16):
    w = (torch.randn(R, C, device="cuda") * 0.02).to(dt)
    for name, zd in (("sint4", None), ("uint4", ZeroPointDomain.PostScale)):
        q = Quantizer(QuantizerConfig(dtype=getattr(QDType, name), zero_point=zd, group_shapes=((1, 128),), scale_dtypes=(torch.float16,)))
        info = q.update(w.shape, default_dtype=dt)
        step = info.steps[0].scale.native(w.device)
        x2d, _ = engine.canonical_groups(w, info.steps[0].tensor_view_shape)
        ms = timeit(lambda: engine.fake_quant_fused(step, x2d))
        print(f"fake_quant fused {dt} {name}: {ms*1e3:.1f} us  {R*C*4/ms/1e6:.0f} GB/s (4 B/param)  (incl. alloc)")
        ms = timeit(lambda: q.quantize(w))
        print(f"  Quantizer.quantize end-to-end: {ms*1e3:.1f} us")
        ratios = [r for it in DynamicRangeCalibConfig(strategy=SearchBasedCalibStrategy.GridSearch, num_grids=100).get_ratios() for r in it]
        for deg in (2.4, 2):
            ms = timeit(lambda: engine.range_search_groups(step, x2d, ratios, deg), iters=5, warm=1)
            print(f"  range search K={len(ratios)} deg={deg}: {ms:.3f} ms  {R*C/ms/1e6:.2f} Gparam/s")
