"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's rotation and manual smoothing stages on a
Llama-layout state dict (deepcompressor/app/llm/quant/rotate.py:25-196, app/llm/quant/smooth.py:23-167), composed
from the rotation and smoothing oracles.  Imported by tests only; the product never imports this module.
Pinned by ``tests/golden/llm_layer_stages.npz`` (generated from the reference by ``oracle/make_golden.py``).

State dicts map parameter names to float32 arrays holding bfloat16 values."""

from __future__ import annotations

import numpy as np

import quant_oracle as qo
import rotate_oracle as ro
import smooth_oracle as so

F32 = np.float32


def _fold(norm_w: np.ndarray, weight: np.ndarray) -> np.ndarray:
    """transform_rms_norm_and_linear (calib/rotate.py:136-162) on a float32 linear: (W.f64 * ln.f64).to(float32)."""
    return (weight.astype(np.float64) * norm_w.astype(np.float64)[None, :]).astype(F32)


def rotate_llm(state: dict[str, np.ndarray], rotation: np.ndarray, num_layers: int, head_channels: int,
               transform_out_proj: bool = True) -> dict[str, np.ndarray]:
    """rotate.py:25-196 for a Llama layout: the linears take part as float32 copies and are cast back to bfloat16
    at the end; norms and the embedding stay bfloat16 throughout."""
    out = {k: v.copy() for k, v in state.items()}
    lin = lambda name: out[name].astype(F32)  # noqa: E731
    work: dict[str, np.ndarray] = {}
    for i in range(num_layers):
        p = f"model.layers.{i}"
        for proj in ("q_proj", "k_proj", "v_proj"):
            work[f"{p}.self_attn.{proj}.weight"] = _fold(out[f"{p}.input_layernorm.weight"], lin(f"{p}.self_attn.{proj}.weight"))
        out[f"{p}.input_layernorm.weight"] = np.ones_like(out[f"{p}.input_layernorm.weight"])
        for proj in ("up_proj", "gate_proj"):
            work[f"{p}.mlp.{proj}.weight"] = _fold(out[f"{p}.post_attention_layernorm.weight"], lin(f"{p}.mlp.{proj}.weight"))
        out[f"{p}.post_attention_layernorm.weight"] = np.ones_like(out[f"{p}.post_attention_layernorm.weight"])
        work[f"{p}.self_attn.o_proj.weight"] = lin(f"{p}.self_attn.o_proj.weight")
        work[f"{p}.mlp.down_proj.weight"] = lin(f"{p}.mlp.down_proj.weight")
    work["lm_head.weight"] = _fold(out["model.norm.weight"], lin("lm_head.weight"))
    out["model.norm.weight"] = np.ones_like(out["model.norm.weight"])
    out["model.embed_tokens.weight"] = ro.rotate_in_channels(out["model.embed_tokens.weight"], "bf16", rotation)
    head_rotation = ro.sylvester(head_channels) / np.sqrt(head_channels)
    for i in range(num_layers):
        p = f"model.layers.{i}"
        for name in (f"{p}.self_attn.q_proj", f"{p}.self_attn.k_proj", f"{p}.self_attn.v_proj", f"{p}.mlp.up_proj",
                     f"{p}.mlp.gate_proj"):
            work[f"{name}.weight"] = ro.rotate_in_channels(work[f"{name}.weight"], "f32", rotation)
        work[f"{p}.self_attn.o_proj.weight"] = ro.rotate_out_channels(work[f"{p}.self_attn.o_proj.weight"], "f32", rotation)
        if transform_out_proj:
            work[f"{p}.self_attn.v_proj.weight"] = ro.rotate_out_channels(work[f"{p}.self_attn.v_proj.weight"], "f32",
                                                                          head_rotation)
            work[f"{p}.self_attn.o_proj.weight"] = ro.rotate_in_channels(work[f"{p}.self_attn.o_proj.weight"], "f32",
                                                                         head_rotation)
        work[f"{p}.mlp.down_proj.weight"] = ro.rotate_out_channels(work[f"{p}.mlp.down_proj.weight"], "f32", rotation)
    work["lm_head.weight"] = ro.rotate_in_channels(work["lm_head.weight"], "f32", rotation)
    for name, w in work.items():
        out[name] = qo.round_to("bf16", w)
    return out


def manual_proj_scale(inputs, weights, alpha: float, beta: float, num_heads: int = 1, num_head_repeats: int = 1):
    """The scale smooth_linear_modules finds with the Manual strategy and AbsMax spans (calib/smooth.py:272-386):
    x span ^ alpha / w span ^ beta, spans shared by the query heads of one key/value head."""
    channels = weights[0].shape[1]
    a = so.channel_metric([x.reshape(-1, channels) for x in inputs], channels, "abs_max")
    b = so.channel_metric(weights, channels, "abs_max")
    if num_heads > 1 and num_head_repeats > 1:
        a = so._share(a, num_heads // num_head_repeats, num_head_repeats, True)
        b = so._share(b, num_heads // num_head_repeats, num_head_repeats, True)
    return so.smooth_scale(a, b, alpha, beta)
