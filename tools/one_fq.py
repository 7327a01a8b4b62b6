"""Run the fused fake-quant kernel a few times on one big tensor (for counter collection)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcompressor_amd import engine
from deepcompressor_amd._lib import lib, ptr, storage_of, stream_of
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig
R, Cn = int(os.environ.get("R", 32768)), int(os.environ.get("C", 8192))
dt = torch.bfloat16
zd = ZeroPointDomain.PostScale if os.environ.get("ZP") else None
w = (torch.randn(R, Cn, device="cuda") * 0.02).to(dt)
q = Quantizer(QuantizerConfig(dtype=QDType.uint4 if zd else QDType.sint4, zero_point=zd, group_shapes=((1, 128),), scale_dtypes=(torch.float16,)))
info = q.update(w.shape, default_dtype=dt)
step = info.steps[0].scale.native(w.device)
x2d, _ = engine.canonical_groups(w, info.steps[0].tensor_view_shape)
G, S = x2d.shape
dq = torch.empty_like(x2d); sc = torch.empty(G, device="cuda"); ze = torch.empty(G, device="cuda")
for _ in range(4):
    lib().dcq_fake_quant_groups(C.byref(step), ptr(x2d), storage_of(dt), G, S, None, None, 0, 0.0, ptr(dq), None, ptr(sc), ptr(ze), None, stream_of(w.device))
torch.cuda.synchronize()
