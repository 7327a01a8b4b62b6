"""Rotation, channel-reorder and smoothing stages of a decoder-only language model
(reference: deepcompressor/app/llm/quant/rotate.py:25-196, app/llm/quant/reorder.py:48-236,
app/llm/quant/smooth.py:23-167).

The drivers walk struct objects (``llm_struct``, or the reference's own -- the attribute names are the same) and call
the building blocks of ``calib.rotate`` / ``calib.smooth``; all arithmetic runs on the device the modules live on.
"""

from __future__ import annotations

import typing as tp
from dataclasses import dataclass

import torch
import torch.nn as nn

from ..calib.config.reorder import SkipBasedChannelOrderConfig
from ..calib.config.rotation import QuantRotationConfig
from ..calib.config.smooth import SmoothTransfomerConfig
from ..calib.reorder import ChannelOrderCalibrator, ChannelReorderer
from ..calib.rotate import (get_rotation_matrix, hadamard_in_channels, rotate_in_channels, rotate_out_channels,
                            transform_norm_and_linear)
from ..calib.smooth import ActivationSmoother, smooth_attention, smooth_linear_modules
from ..data.cache import IOTensorsCache
from ..quantizer.processor import Quantizer
from ..data.common import TensorType
from .llm import (ActivationEntry, LinearEntry, LlmActivationQuantizerConfig, LlmWeightQuantizerConfig,
                  quantize_layer_activations, quantize_layer_weights)
from .llm_struct import LlmModelStruct, LlmTransformerBlockStruct

__all__ = ["LlmQuantConfig", "rotate_llm", "reorder_llm_layer", "smooth_llm_layer", "quantize_llm_layer_weights",
           "quantize_llm_layer_activations"]


@dataclass
class LlmQuantConfig:
    """The quantization settings the stages read (app/llm/quant/config.py:31-120, without the calibration data
    loader, which belongs to the caller here)."""

    wgts: LlmWeightQuantizerConfig | None = None
    ipts: LlmActivationQuantizerConfig | None = None
    opts: LlmActivationQuantizerConfig | None = None
    rotation: QuantRotationConfig | None = None
    reorder: SkipBasedChannelOrderConfig | None = None
    smooth: SmoothTransfomerConfig | None = None
    develop_dtype: torch.dtype = torch.float32

    def __post_init__(self) -> None:
        if self.smooth is not None and not self.smooth.enabled_proj and not self.smooth.enabled_attn:
            self.smooth = None
        if self.rotation is not None and self.reorder is not None:
            self.reorder.skips.append("residual")
            if self.rotation.transforms:
                self.reorder.skips.extend(self.rotation.transforms)
            self.reorder.skips = sorted(set(self.reorder.skips))

    @staticmethod
    def _enabled(q) -> bool:
        return q is not None and q.is_enabled()

    @property
    def enabled_wgts(self) -> bool:
        return self._enabled(self.wgts)

    @property
    def enabled_ipts(self) -> bool:
        return self._enabled(self.ipts)

    @property
    def enabled_opts(self) -> bool:
        return self._enabled(self.opts)

    @property
    def enabled_smooth(self) -> bool:
        return self.smooth is not None

    @property
    def enabled_rotation(self) -> bool:
        return self.rotation is not None


@torch.inference_mode()
def rotate_llm(model: nn.Module | LlmModelStruct, /, config: QuantRotationConfig,
               rotation: torch.Tensor | None = None) -> torch.Tensor:
    """Fold the norms into the linears behind them, then rotate the residual stream: embeddings and every linear that
    reads it on the input side, every linear that writes it on the output side; with ``out_proj`` among the
    transforms the value / output projections also get a per-head rotation, with ``down_proj`` the down
    projections an online Hadamard transform.  The linears are rotated in float32 and cast back, as the reference
    does (rotate.py:25-196).  Returns the rotation matrix."""
    if not hasattr(model, "backbone_struct"):
        model = LlmModelStruct.construct(model)
    head = model.head if model.backbone_struct.proj_out is None else model.backbone_struct.proj_out
    assert head is not None, "rotation needs the projection behind the final norm (the language-model head)"
    if model.backbone_struct.proj_out is None and head.weight is model.backbone_struct.embed_tokens.weight:
        # the embedding and the head are rotated one after the other: a shared tensor would be rotated twice
        raise ValueError("rotation of a model with tied input / output embeddings: untie the weights first")
    linears = [m for m in model.module.modules() if isinstance(m, nn.Linear)]
    dtypes = [m.weight.dtype for m in linears]
    for linear in linears:
        linear.to(dtype=torch.float32)
    for block in model.iter_transformer_block_structs():
        assert not block.post_attn_norms, "Rotation is only supported for models without post-attention norms."
        assert not block.post_ffn_norm, "Rotation is only supported for models without post-FFN norms."
    backbone = model.backbone_struct
    layers = backbone.layer_structs
    if backbone.norm_in is None:
        if backbone.proj_in is None:
            prev_modules, prev_dims = [backbone.embed_tokens], 1
            if backbone.embed_positions is not None:
                prev_modules.append(backbone.embed_positions)
        elif backbone.embed_positions is None:
            prev_modules, prev_dims = [backbone.proj_in], 0
        else:
            prev_modules, prev_dims = [backbone.proj_in, backbone.embed_positions], [0, 1]
    else:
        prev_modules, prev_dims = [backbone.norm_in], 0
    for layer in layers:
        transform_norm_and_linear(parent=layer.module, norm_name=layer.pre_attn_norm_rname,
                                  next_modules=layer.attn_struct.qkv_proj, prev_modules=prev_modules,
                                  prev_out_channels_dims=prev_dims)
        prev_modules, prev_dims = [layer.attn_struct.out_proj], 0
        ffn = layer.ffn_struct
        transform_norm_and_linear(parent=layer.module, norm_name=layer.pre_ffn_norm_rname,
                                  next_modules=ffn.up_projs + ([ffn.moe_gate] if ffn.moe_gate is not None else []),
                                  prev_modules=prev_modules, prev_out_channels_dims=prev_dims)
        prev_modules, prev_dims = ffn.down_projs, 0
    transform_norm_and_linear(parent=backbone.module, norm_name=backbone.norm_out_rname, next_modules=[head],
                              prev_modules=prev_modules, prev_out_channels_dims=prev_dims)
    num_channels = getattr(backbone, "num_channels", None) or backbone.config.num_channels
    if rotation is None:
        rotation = get_rotation_matrix(num_channels, random=config.random)
    device = head.weight.device
    rotation = rotation.to(device)
    if backbone.proj_in is None:
        rotate_in_channels(backbone.embed_tokens.weight, rotation=rotation)
    else:
        rotate_out_channels(backbone.proj_in.weight, rotation=rotation, bias=backbone.proj_in.bias)
    if backbone.embed_positions is not None:
        rotate_in_channels(backbone.embed_positions.weight, rotation=rotation)
    num_head_channels = (model.num_head_channels if hasattr(model, "num_head_channels")
                         else model.config.num_head_channels)
    head_rotation = get_rotation_matrix(num_head_channels, random=config.random).to(device)
    down_proj = []
    for layer in layers:
        attn, ffn = layer.attn_struct, layer.ffn_struct
        for proj in attn.qkv_proj:
            rotate_in_channels(proj.weight, rotation=rotation)
        rotate_out_channels(attn.out_proj.weight, rotation=rotation, bias=attn.out_proj.bias)
        if attn.out_proj_key in config.transforms:
            rotate_out_channels(attn.v_proj.weight, rotation=head_rotation, bias=attn.v_proj.bias)
            rotate_in_channels(attn.o_proj.weight, rotation=head_rotation)
        for fc in ffn.up_projs:
            rotate_in_channels(fc.weight, rotation=rotation)
        if ffn.moe_gate is not None:
            rotate_in_channels(ffn.moe_gate.weight, rotation=rotation)
        for fc in ffn.down_projs:
            rotate_out_channels(fc.weight, rotation=rotation, bias=fc.bias)
        if ffn.down_proj_key in config.transforms:
            down_proj.extend(ffn.down_projs)
    if backbone.proj_out is not None:
        rotate_in_channels(backbone.proj_out.weight, rotation=rotation)
        rotate_out_channels(backbone.proj_out.weight, rotation=rotation, bias=backbone.proj_out.bias)
    if down_proj:
        hadamard_in_channels(down_proj)
    rotate_in_channels((backbone.proj_out if backbone.proj_out is not None else model.head).weight, rotation=rotation)
    for dtype, linear in zip(dtypes, linears, strict=True):
        linear.to(dtype=dtype)
    return rotation.cpu()  # as the reference does (rotate.py:196): the matrix is stage output, kept in host memory


@torch.inference_mode()
def smooth_llm_layer(layer: LlmTransformerBlockStruct, config: LlmQuantConfig, smooth_cache: dict[str, torch.Tensor],
                     layer_cache: dict[str, IOTensorsCache] | None = None,
                     layer_kwargs: dict[str, tp.Any] | None = None) -> None:
    """Smooth one decoder layer (smooth.py:23-167): the query-key product, then the inputs of the q/k/v projections
    (against the norm in front), of the output projection (against v_proj), of the up projections (against the norm
    in front) and of every down projection (against its up projection).  Scales found are stored in
    ``smooth_cache``; scales already there are applied as they are."""
    layer_cache, layer_kwargs = layer_cache or {}, layer_kwargs or {}
    attn, ffn = layer.attn_struct, layer.ffn_struct
    dev_dtype = config.develop_dtype

    def needs_quant(key: str) -> bool:
        return ((config.enabled_ipts and config.ipts.is_enabled_for(key))
                or (config.enabled_wgts and config.wgts.is_enabled_for(key)))

    def proj_enabled(key: str) -> bool:
        return config.smooth.enabled_proj and config.smooth.proj.is_enabled_for(key) and needs_quant(key)

    def io(name: str, outputs: bool = False):
        if not layer_cache:
            return None
        return layer_cache[name].outputs if outputs else layer_cache[name].inputs

    if (config.smooth.enabled_attn and config.enabled_opts
            and (config.opts.is_enabled_for(attn.q_key) or config.opts.is_enabled_for(attn.k_key))):
        cache_key = f"{attn.name}.{attn.k_rkey}"
        smooth_cache[cache_key] = smooth_attention(
            k_proj=attn.k_proj, q_proj=attn.q_proj, scale=smooth_cache.get(cache_key), config=config.smooth.attn,
            query_quantizer=Quantizer(config.opts, channels_dim=-1, key=attn.q_key),
            key_quantizer=Quantizer(config.opts, channels_dim=-1, key=attn.k_key),
            queries=io(attn.q_name, True), keys=io(attn.k_name, True), attn_q=attn.q, attn_k=attn.k,
            eval_inputs=io(attn.name), eval_module=attn.module,
            eval_kwargs=attn.filter_kwargs(layer_kwargs), num_heads=attn.config.num_query_heads,
            num_head_repeats=attn.config.num_head_repeats, with_rope=attn.config.with_rope, develop_dtype=dev_dtype)
    if proj_enabled(attn.qkv_proj_key):
        cache_key = attn.v_proj_name
        smooth_cache[cache_key] = smooth_linear_modules(
            attn.parent.pre_attn_norm, attn.qkv_proj, scale=smooth_cache.get(cache_key), config=config.smooth.proj,
            weight_quantizer=Quantizer(config.wgts, key=attn.qkv_proj_key),
            input_quantizer=Quantizer(config.ipts, channels_dim=-1, key=attn.qkv_proj_key),
            inputs=io(attn.q_proj_name), eval_inputs=io(attn.name), eval_module=attn.module,
            eval_kwargs=attn.filter_kwargs(layer_kwargs), develop_dtype=dev_dtype)
        if not attn.parent.pre_attn_norm:
            ActivationSmoother(smooth_cache[cache_key], channels_dim=-1).as_hook().register(attn.qkv_proj)
    if proj_enabled(attn.out_proj_key):
        cache_key = attn.o_proj_name
        smooth_cache[cache_key] = smooth_linear_modules(
            None if attn.config.linear_attn else attn.v_proj, attn.o_proj, scale=smooth_cache.get(cache_key),
            config=config.smooth.proj, weight_quantizer=Quantizer(config.wgts, key=attn.out_proj_key),
            input_quantizer=Quantizer(config.ipts, channels_dim=-1, key=attn.out_proj_key), inputs=io(cache_key),
            eval_inputs=io(cache_key), eval_module=attn.o_proj, num_heads=attn.config.num_query_heads,
            num_head_repeats=attn.config.num_head_repeats, develop_dtype=dev_dtype)
        if attn.config.linear_attn:
            ActivationSmoother(smooth_cache[cache_key], channels_dim=-1).as_hook().register(attn.o_proj)
    num_experts = ffn.config.num_experts
    if proj_enabled(ffn.up_proj_key):
        cache_key = ffn.name
        smooth_cache[cache_key] = smooth_linear_modules(
            ffn.parent.pre_ffn_norm, ffn.up_projs, scale=smooth_cache.get(cache_key), config=config.smooth.proj,
            weight_quantizer=Quantizer(config.wgts, key=ffn.up_proj_key),
            input_quantizer=Quantizer(config.ipts, channels_dim=-1, key=ffn.up_proj_key), inputs=io(ffn.name),
            eval_inputs=io(ffn.name), eval_module=ffn.module,
            extra_modules=[ffn.moe_gate] if num_experts > 1 else None, develop_dtype=dev_dtype)
        if not ffn.parent.pre_ffn_norm:
            hook = ActivationSmoother(smooth_cache[cache_key], channels_dim=-1).as_hook().register(ffn.up_projs)
            if num_experts > 1:
                hook.register(ffn.moe_gate)
    if proj_enabled(ffn.down_proj_key):
        for expert_idx in range(num_experts):
            cache_key = ffn.down_proj_names[expert_idx]
            smooth_cache[cache_key] = smooth_linear_modules(
                ffn.up_projs[expert_idx], ffn.down_projs[expert_idx], scale=smooth_cache.get(cache_key),
                config=config.smooth.proj, weight_quantizer=Quantizer(config.wgts, key=ffn.down_proj_key),
                input_quantizer=Quantizer(config.ipts, channels_dim=-1, key=ffn.down_proj_key), inputs=io(cache_key),
                eval_inputs=io(cache_key), eval_module=ffn.down_projs[expert_idx], develop_dtype=dev_dtype)


@torch.inference_mode()
def reorder_llm_layer(layer: LlmTransformerBlockStruct, config: LlmQuantConfig, reorder_cache: dict[str, torch.Tensor],
                      residual_calibrator: ChannelOrderCalibrator | None = None,
                      layer_cache: dict[str, IOTensorsCache] | None = None,
                      layer_kwargs: dict[str, tp.Any] | None = None) -> ChannelOrderCalibrator | None:
    """Find (or take from ``reorder_cache``) and apply the input-channel orders inside one decoder layer
    (reorder.py:48-236): the output projection (its order, reduced to the key/value heads, also permutes v_proj's
    rows) and every down projection (permuting the rows of its up projections); with ``dynamic`` orders also the
    q/k/v and up projections, whose inputs are then permuted at run time by a hook.  Returns the calibrator that
    collects the residual stream's channel metrics, or None once it is of no use."""
    layer_cache, layer_kwargs = layer_cache or {}, layer_kwargs or {}
    attn, ffn = layer.attn_struct, layer.ffn_struct
    qkv_proj, out_proj = attn.qkv_proj, attn.out_proj
    num_heads, num_head_repeats = attn.config.num_query_heads, attn.config.num_head_repeats

    def inputs(name: str):
        return layer_cache[name].inputs if layer_cache else None

    def keep(index: torch.Tensor | None, channels: int) -> torch.Tensor:
        # a search that the given order wins hands back None (the reference's driver then fails on ``None.to``,
        # reorder.py:104, 133, 179, 208); here that is stored as the identity order
        return (torch.arange(channels) if index is None else index).to(device=torch.device("cpu"))

    def calibrator(key: str, **heads) -> ChannelOrderCalibrator:
        return ChannelOrderCalibrator(config=config.reorder, weight_quantizer=Quantizer(config.wgts, key=key),
                                      input_quantizer=Quantizer(config.ipts, channels_dim=-1, key=key),
                                      develop_dtype=config.develop_dtype, **heads)

    if config.reorder.dynamic and config.reorder.is_enabled_for(attn.qkv_proj_key):
        cache_key = attn.name
        if cache_key not in reorder_cache:
            index = calibrator(attn.qkv_proj_key).calibrate(
                x_wgts=[m.weight for m in qkv_proj], x_acts=inputs(attn.v_proj_name), x_mods=qkv_proj,
                eval_inputs=inputs(cache_key), eval_module=attn.module, eval_kwargs=attn.filter_kwargs(layer_kwargs),
                reorder_wgts=[(m.weight, 1) for m in qkv_proj], reorder_ipt_mods=[(attn.module, -1, None)],
                reorder_opt_mods=[])
            reorder_cache[cache_key] = keep(index, qkv_proj[0].weight.shape[1])
        index = reorder_cache[cache_key]
        for proj in qkv_proj:
            proj.weight.data = proj.weight.data.index_select(1, index.to(proj.weight.device))
        ChannelReorderer(index, channels_dim=-1).as_hook().register(attn.module)
    if config.reorder.is_enabled_for(attn.out_proj_key):
        cache_key = attn.out_proj_name
        if cache_key not in reorder_cache:
            index = calibrator(attn.out_proj_key, num_heads=num_heads, num_head_repeats=num_head_repeats).calibrate(
                x_wgts=[out_proj.weight], x_acts=inputs(cache_key), x_mods=[out_proj], eval_inputs=inputs(cache_key),
                eval_module=out_proj, reorder_wgts=[(out_proj.weight, 1)], reorder_ipt_mods=[(out_proj, -1, None)],
                reorder_opt_mods=[])
            reorder_cache[cache_key] = keep(index, out_proj.weight.shape[1])
        index = reorder_cache[cache_key].to(out_proj.weight.device)
        out_proj.weight.data = out_proj.weight.data.index_select(1, index)
        v_proj = qkv_proj[2]
        if num_heads > 1 and num_head_repeats > 1:
            # the order inside the first query head of every group, as an order of the key/value head's channels
            num_channels = index.numel()
            head_channels = num_channels // num_heads
            index = index.view(num_heads, head_channels)
            index = index - torch.arange(0, num_channels, head_channels, device=index.device).view(num_heads, 1)
            num_v_heads = num_heads // num_head_repeats
            index = index.view(num_v_heads, num_head_repeats, head_channels)[:, 0, :]
            index = index + torch.arange(0, num_channels // num_head_repeats, head_channels,
                                         device=index.device).view(num_v_heads, 1)
            index = index.reshape(-1)
        v_proj.weight.data = v_proj.weight.data.index_select(0, index.to(v_proj.weight.device))
        if v_proj.bias is not None:
            v_proj.bias.data = v_proj.bias.data[index.to(v_proj.bias.device)].contiguous()
    num_experts = ffn.config.num_experts
    up_proj, down_proj = ffn.up_projs, ffn.down_projs
    if config.reorder.dynamic and config.reorder.is_enabled_for(ffn.up_proj_key):
        cache_key = ffn.name
        if cache_key not in reorder_cache:
            index = calibrator(ffn.up_proj_key).calibrate(
                x_wgts=[m.weight for m in up_proj], x_acts=inputs(cache_key), x_mods=up_proj,
                eval_inputs=inputs(cache_key), eval_module=ffn.module, reorder_wgts=[(m.weight, 1) for m in up_proj],
                reorder_ipt_mods=[(ffn.module, -1, None)], reorder_opt_mods=[])
            reorder_cache[cache_key] = keep(index, up_proj[0].weight.shape[1])
        index = reorder_cache[cache_key]
        for fc in up_proj:
            fc.weight.data = fc.weight.data.index_select(1, index.to(fc.weight.device))
        if ffn.moe_gate is not None:
            ffn.moe_gate.weight.data = ffn.moe_gate.weight.data.index_select(1, index.to(ffn.moe_gate.weight.device))
        ChannelReorderer(index, channels_dim=-1).as_hook().register(ffn.module)
    if config.reorder.is_enabled_for(ffn.down_proj_key):
        for expert_idx, (fc2_name, fc2) in enumerate(zip(ffn.down_proj_names, down_proj, strict=True)):
            cache_key = fc2_name
            if cache_key not in reorder_cache:
                index = calibrator(ffn.down_proj_key).calibrate(
                    x_wgts=[fc2.weight], x_acts=inputs(cache_key), x_mods=[fc2], eval_inputs=inputs(cache_key),
                    eval_module=fc2, reorder_wgts=[(fc2.weight, 1)], reorder_ipt_mods=[(fc2, -1, None)],
                    reorder_opt_mods=[])
                reorder_cache[cache_key] = keep(index, fc2.weight.shape[1])
            index = reorder_cache[cache_key].to(fc2.weight.device)
            fc2.weight.data = fc2.weight.data.index_select(1, index)
            for fc1 in up_proj[expert_idx::num_experts]:
                fc1.weight.data = fc1.weight.data.index_select(0, index.to(fc1.weight.device))
                if fc1.bias is not None:
                    fc1.bias.data = fc1.bias.data[index.to(fc1.bias.device)].contiguous()
    if residual_calibrator is not None and (config.reorder.dynamic
                                            or not config.reorder.is_enabled_for(attn.qkv_proj_key)
                                            or not config.reorder.is_enabled_for(ffn.up_proj_key)):
        residual_calibrator = None
    if residual_calibrator is not None and "residual" not in reorder_cache:
        residual_calibrator.update_channel_metrics(weights=[m.weight for m in qkv_proj], inputs=inputs(attn.v_proj_name))
        for expert_idx in range(num_experts):
            residual_calibrator.update_channel_metrics(weights=[m.weight for m in up_proj[expert_idx::num_experts]],
                                                       inputs=inputs(ffn.up_proj_names[expert_idx]))
    return residual_calibrator


@torch.inference_mode()
def quantize_llm_layer_weights(layer: LlmTransformerBlockStruct, config: LlmQuantConfig,
                               quantizer_state_dict: dict[str, tp.Any],
                               layer_cache: dict[str, IOTensorsCache] | None = None,
                               layer_kwargs: dict[str, tp.Any] | None = None,
                               return_with_scale_state_dict: bool = False) -> dict[str, torch.Tensor | float | None]:
    """The weight stage over a layer struct (app/llm/quant/weight.py:24-105): q_proj / k_proj ranges are scored on
    their whole attention block, every other linear on itself."""
    layer_kwargs = layer_kwargs or {}
    entries = []
    for key, name, module, parent, field_name in layer.named_key_modules():
        if field_name in ("q_proj", "k_proj"):
            entries.append(LinearEntry(key, name, module, eval_name=parent.name, eval_module=parent.module,
                                       eval_kwargs=parent.filter_kwargs(layer_kwargs)))
        else:
            entries.append(LinearEntry(key, name, module))
    return quantize_layer_weights(entries, config.wgts, quantizer_state_dict, layer_cache=layer_cache,
                                  return_with_scale_state_dict=return_with_scale_state_dict,
                                  develop_dtype=config.develop_dtype)


@torch.inference_mode()
def quantize_llm_layer_activations(layer: LlmTransformerBlockStruct, config: LlmQuantConfig,
                                   quantizer_state_dict: dict[str, tp.Any],
                                   layer_cache: dict[str, IOTensorsCache] | None = None,
                                   layer_kwargs: dict[str, tp.Any] | None = None,
                                   orig_state_dict: dict[str, torch.Tensor] | None = None):
    """The activation stage over a layer struct (app/llm/quant/activation.py:25-192): the shared input of q/k/v (cached
    under v_proj, scored on the attention block), the queries / keys / values leaving ``attn.q`` / ``attn.k`` /
    ``attn.v`` (scored on the attention block), the inputs of the output projection, of the up projections (scored
    on the feed-forward block) and of every down projection.  Returns ``{cache key: (quantizer, hooks)}``."""
    layer_kwargs, orig = layer_kwargs or {}, orig_state_dict or {}
    attn, ffn = layer.attn_struct, layer.ffn_struct
    attn_kwargs = attn.filter_kwargs(layer_kwargs)
    In, Out = TensorType.Inputs, TensorType.Outputs

    def originals(names, modules):
        return [(m.weight, orig[f"{n}.weight"]) for n, m in zip(names, modules, strict=True)] if orig else None

    attn_orig = originals(attn.qkv_proj_names + [attn.out_proj_name], attn.qkv_proj + [attn.out_proj])
    entries = [ActivationEntry(attn.qkv_proj_key, In, attn.qkv_proj, attn.v_proj_name, attn.module, attn.name,
                               attn_kwargs, attn_orig)]
    doubled = (attn_orig + attn_orig) if attn_orig else None
    for idx, key in enumerate((attn.q_key, attn.k_key, attn.v_key)):
        module, name = getattr(attn, "qkv"[idx]), getattr(attn, f"{'qkv'[idx]}_name")
        entries.append(ActivationEntry(key, Out, [module], name, attn.module, attn.name, attn_kwargs,
                                       doubled[idx:idx + 4] if doubled else None))
    entries.append(ActivationEntry(attn.out_proj_key, In, [attn.out_proj], attn.out_proj_name, attn.out_proj,
                                   attn.out_proj_name, None, originals([attn.out_proj_name], [attn.out_proj])))
    num_experts = ffn.config.num_experts
    for expert_idx in range(num_experts):
        ups, up_names = ffn.up_projs[expert_idx::num_experts], ffn.up_proj_names[expert_idx::num_experts]
        entries.append(ActivationEntry(ffn.up_proj_key, In, ups, up_names[0], ffn.experts[expert_idx], up_names[0], None,
                                       originals(up_names, ups)))
        down, down_name = ffn.down_projs[expert_idx], ffn.down_proj_names[expert_idx]
        entries.append(ActivationEntry(ffn.down_proj_key, In, [down], down_name, down, down_name, None,
                                       originals([down_name], [down])))
    return quantize_layer_activations(entries, config.ipts if config.enabled_ipts else None,
                                      config.opts if config.enabled_opts else None, quantizer_state_dict,
                                      layer_cache=layer_cache, develop_dtype=config.develop_dtype)
