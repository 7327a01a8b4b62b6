"""Query / key smoothing search on the GPU (HIP channel spans, channel scaling and fake quantization through the
C ABI; the attention scores module is the caller's torch module) against fixtures captured from the reference."""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import to_np
from test_oracle_smooth_attn import G, config_of, ulps

from deepcompressor_amd.calib.smooth import smooth_attention
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig

pytestmark = pytest.mark.gpu


class ToyScores(nn.Module):
    def __init__(self, channels, heads, kv_heads, head_dim):
        super().__init__()
        self.heads, self.kv_heads, self.head_dim = heads, kv_heads, head_dim
        self.q_proj = nn.Linear(channels, heads * head_dim, bias=False)
        self.k_proj = nn.Linear(channels, kv_heads * head_dim, bias=False)
        self.attn_q, self.attn_k = nn.Identity(), nn.Identity()

    def forward(self, x):
        b, t, _ = x.shape
        q = self.attn_q(self.q_proj(x)).view(b, t, self.heads, self.head_dim).transpose(1, 2)
        k = self.attn_k(self.k_proj(x)).view(b, t, self.kv_heads, self.head_dim).transpose(1, 2)
        k = k.repeat_interleave(self.heads // self.kv_heads, dim=1)
        return q @ k.transpose(-1, -2)


@pytest.mark.parametrize("tag", sorted(G.meta))
def test_attention_search_fixture(tag):
    m = G.meta[tag]
    dev = torch.device("cuda")
    mod = ToyScores(m["channels"], m["heads"], m["kv_heads"], m["head_dim"]).to(dev)
    mod.q_proj.weight.data = torch.from_numpy(G[f"{tag}|wq"]).to(dev)
    mod.k_proj.weight.data = torch.from_numpy(G[f"{tag}|wk"]).to(dev)
    xs = [torch.from_numpy(G[f"{tag}|x{i}"]).to(dev) for i in range(3)]
    with torch.no_grad():
        queries, keys = [mod.q_proj(x) for x in xs], [mod.k_proj(x) for x in xs]
    cache = lambda data: TensorsCache(TensorCache(list(data), channels_dim=-1, reshape=LinearReshapeFn(),  # noqa: E731
                                                  num_cached=3, num_total=3, num_samples=6, orig_device=dev))
    kq = Quantizer(QuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale,
                                   group_shapes=((1, m["head_dim"]),), scale_dtypes=(torch.float16,)), channels_dim=-1,
                   key="k")
    qq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),),
                                   scale_dtypes=(torch.float16,)), channels_dim=-1, key="q") if m["quant_q"] else None
    history = []
    import deepcompressor_amd.calib.smooth as smooth_mod

    plain_tell = smooth_mod.SmoothCalibrator.tell

    def tell(self, error):
        history.append((float(error[0]), self.candidate.clone()))
        return plain_tell(self, error)

    smooth_mod.SmoothCalibrator.tell = tell
    try:
        reps = m["heads"] // m["kv_heads"]
        best = smooth_attention(q_proj=mod.q_proj, k_proj=mod.k_proj, scale=None, config=config_of(m),
                                query_quantizer=qq, key_quantizer=kq, queries=cache(queries), keys=cache(keys),
                                attn_q=mod.attn_q, attn_k=mod.attn_k, eval_inputs=cache(xs), eval_module=mod,
                                eval_kwargs={}, num_heads=m["heads"], num_head_repeats=reps, with_rope=m["with_rope"])
    finally:
        smooth_mod.SmoothCalibrator.tell = plain_tell
    errors = np.array([e for e, _ in history])
    scales = np.stack([to_np(s) for _, s in history])
    assert scales.shape == G[f"{tag}|scales"].shape
    assert ulps(scales, G[f"{tag}|scales"]).max() <= 4
    np.testing.assert_allclose(errors, G[f"{tag}|errors"], rtol=1e-2)
    assert int(np.argmin(errors)) == m["best_index"]
    assert best.device.type == "cpu" and ulps(best.numpy(), G[f"{tag}|best_scale"]).max() <= 4
    # the projections end up scaled by the winner: q rows up, k rows down
    np.testing.assert_allclose(to_np(mod.q_proj.weight.data), G[f"{tag}|wq_out"], rtol=1e-6)
    np.testing.assert_allclose(to_np(mod.k_proj.weight.data), G[f"{tag}|wk_out"], rtol=1e-6)
