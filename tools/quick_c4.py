"""Times the smoothing search of the Hiera-L fc1 case (39 candidates, 4096 tokens, float32 W8A8) with the fused
output scorer and with the module evaluated candidate by candidate; prints the per-stage times of the fused path."""
import sys
import time

import torch
from torch import nn

sys.path.insert(0, ".")
from deepcompressor_amd import engine  # noqa: E402
from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,  # noqa: E402
                                              SearchBasedCalibStrategy, SmoothCalibConfig, SmoothSpanMode)
from deepcompressor_amd.calib.smooth import SmoothLinearCalibrator  # noqa: E402
from deepcompressor_amd.data.cache import TensorCache, TensorsCache  # noqa: E402
from deepcompressor_amd.data.dtype import QDType  # noqa: E402
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn  # noqa: E402
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig  # noqa: E402

device = torch.device("cuda")
g = torch.Generator(device=device).manual_seed(99)
rand = lambda *shape: torch.randn(*shape, generator=g, device=device)  # noqa: E731
fc1 = nn.Linear(1152, 4608, bias=True, device=device, dtype=torch.float32)
fc1.weight.data = rand(4608, 1152) * 0.03
gains = torch.exp(rand(1152) * 1.2)
xs = [rand(1, 1024, 1152) * gains for _ in range(4)]
cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=4, num_total=4,
                                 num_samples=4, orig_device=device))
w8 = QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(None,))
wq, xq = Quantizer(w8, key="w"), Quantizer(w8, channels_dim=-1, key="x")
scfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=20,
                         strategy=SearchBasedCalibStrategy.GridSearch, objective=SearchBasedCalibObjective.OutputsError,
                         granularity=SearchBasedCalibGranularity.Layer)


def run(fused):
    cal = SmoothLinearCalibrator(scfg, wq, xq)
    if not fused:
        cal._all_candidates_linear_outputs_error = lambda *a, **k: None
    best = cal.calibrate(x_wgts=[fc1.weight], x_acts=cache, x_mods=[fc1], eval_inputs=cache, eval_module=fc1,
                         eval_kwargs={})
    return best, torch.stack([e[0].reshape(()) for e in cal.error_history])


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


bf, ef = run(True)
bm, em = run(False)
print("max rel error difference", float((ef / em - 1).abs().max()), "same best", bool(torch.equal(bf, bm)),
      "argmin", int(ef.argmin()), int(em.argmin()))
print("fused search ms", wall(lambda: run(True), 5), "module search ms", wall(lambda: run(False), 3))
# stages of the fused path
cal = SmoothLinearCalibrator(scfg, wq, xq)
cal.reset(x_wgts=[fc1.weight], x_acts=cache)
scales = cal._candidate_table().float()
w = fc1.weight.data
def table():
    cal._scales = None
    return cal._candidate_table()


print("candidate scales ms", wall(table, 5))
print("w codes ms", wall(lambda: wq.smooth_candidates(w, scales, divide_first=False, channels_dim=None,
                                                      default_dtype=w.dtype, as_codes=True), 5))
x = xs[0]
print("x codes ms (one batch)", wall(lambda: xq.smooth_candidates(x, scales, divide_first=True, channels_dim=-1,
                                                                  default_dtype=x.dtype, as_codes=True), 5))
wc, ws = wq.smooth_candidates(w, scales, divide_first=False, channels_dim=None, default_dtype=w.dtype, as_codes=True)
xc, xsc = xq.smooth_candidates(x, scales, divide_first=True, channels_dim=-1, default_dtype=x.dtype, as_codes=True)
y0 = fc1(x).reshape(-1, 4608)
t = wall(lambda: engine.linear_candidate_errors(xc, xsc, wc, ws, fc1.bias, y0), 10)
print("gemm + epilogue ms (one batch, 39 candidates)", t, "Top/s", 2 * 39 * 1024 * 4608 * 1152 / t / 1e9)
