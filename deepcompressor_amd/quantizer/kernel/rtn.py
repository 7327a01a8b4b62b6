"""Round-to-nearest quantization kernel (reference: deepcompressor/quantizer/kernel/rtn.py:15-117), as HIP."""

from __future__ import annotations

import torch

from ... import engine
from ...data.dtype import QuantDataType
from ...data.range import QuantRange
from ...data.zero import ZeroPointDomain
from ..config.kernel import BaseQuantKernel

__all__ = ["QuantRtnKernel", "rtn_quantize"]


def rtn_quantize(tensor: torch.Tensor, *, view_shape: torch.Size, quant_dtype: QuantDataType,
                 zero_domain: ZeroPointDomain | None, scale: torch.Tensor, zero: torch.Tensor,
                 quant_range: QuantRange | None = None, round_delta: torch.Tensor | None = None) -> torch.Tensor:
    """codes = clamp(round((x [+ z]) / s [+ z])) for a float32 working tensor; returns shape ``view_shape``.
    With ``round_delta`` (shaped like the view, or broadcastable to it): clamp(floor(.) + round_delta)."""
    view_shape = torch.Size(tensor.view(view_shape).shape)
    x2d, restore = engine.canonical_groups(tensor, view_shape)
    G = x2d.shape[0]
    qr = QuantRange.construct(quant_dtype, has_zero_point=zero_domain is not None, quant_range=quant_range)
    step = engine.make_step(quant_dtype=quant_dtype, zero_domain=zero_domain, tensor_quant_range=qr,
                            quant_range=quant_range, scale_dtype=torch.float32, zero_dtype=torch.float32,
                            range_bound=None, q_span=1.0, device=tensor.device)
    s = scale.to(torch.float32).expand([v if i % 2 == 0 else 1 for i, v in enumerate(view_shape)]).reshape(G)
    z = None
    if zero_domain is not None:
        z = torch.as_tensor(zero, dtype=torch.float32, device=tensor.device)
        z = z.expand([v if i % 2 == 0 else 1 for i, v in enumerate(view_shape)]).reshape(G).contiguous()
    delta = None
    if round_delta is not None:
        assert not quant_dtype.is_float_point, "round_delta is not supported for float quantization"
        delta, _ = engine.canonical_groups(round_delta.to(torch.float32).expand(view_shape).contiguous(), view_shape)
    _, codes = engine.quantize_groups(step, x2d, s.contiguous(), z, want_dequant=False, want_codes=True,
                                      round_delta=delta)
    return restore(codes).view(view_shape)


class QuantRtnKernel(BaseQuantKernel):
    def quantize(self, tensor: torch.Tensor, *, view_shape: torch.Size, quant_dtype: QuantDataType,
                 zero_domain: ZeroPointDomain | None, scale: torch.Tensor, zero: torch.Tensor,
                 quant_range: QuantRange | None = None, round_delta: torch.Tensor | None = None,
                 **kwargs) -> torch.Tensor:
        return rtn_quantize(tensor, view_shape=view_shape, quant_dtype=quant_dtype, zero_domain=zero_domain,
                            scale=scale, zero=zero, quant_range=quant_range, round_delta=round_delta)
