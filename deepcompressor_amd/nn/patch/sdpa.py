"""Attention product as a module (reference: deepcompressor/nn/patch/sdpa.py:12-26): a hookable stand-in for
``torch.nn.functional.scaled_dot_product_attention`` so that the queries / keys / values entering it can be quantized."""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["ScaleDotProductAttention"]


class ScaleDotProductAttention(nn.Module):
    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                attn_mask: torch.Tensor | None = None, dropout_p: float = 0.0, is_causal: bool = False,
                scale: float | None = None) -> torch.Tensor:
        return F.scaled_dot_product_attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p,
                                              is_causal=is_causal, scale=scale)
