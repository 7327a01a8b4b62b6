"""Per-layer weight quantization stage and its checkpoint files
(reference: deepcompressor/app/llm/quant/quantizer/{config,quantizer}.py, app/llm/quant/weight.py:24-105,
app/llm/ptq.py:228-266).

The reference walks a transformer block through its module-struct classes; here a layer is a plain list of
:class:`LinearEntry` (key, name, module, evaluation module), so any caller that can name its linears can run the
stage.  What it leaves behind is laid out as the reference's files are:

* ``wgts.pt``  -- ``{module_name: Quantizer.state_dict()}`` (the calibrated ranges; reloadable),
* ``scale.pt`` -- ``{param}.scale.{level}`` for every scale level across progressive steps, and ``{param}.zero``
  (zero point in the integer domain) or ``{param}.scaled_zero`` (zero point after scaling),
* ``model.pt`` -- the dequantized weights (``state_dict`` of the module).
"""

from __future__ import annotations

import os
import typing as tp
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from ..calib.config.range import SkipBasedDynamicRangeCalibConfig
from ..calib.range import calibrate_dynamic_range
from ..data.cache import IOTensorsCache, TensorsCache
from ..data.common import TensorType
from ..data.range import DynamicRange
from ..data.zero import ZeroPointDomain
from ..quantizer.config.base import ProgressiveQuantizerConfig
from ..quantizer.kernel.gptq import QuantGptqConfig
from ..quantizer.processor import Quantizer
from ..utils.config import SkipBasedConfig, configclass

__all__ = ["LlmQuantizerConfig", "LlmWeightQuantizerConfig", "LlmActivationQuantizerConfig", "LlmQuantizer",
           "LlmWeightQuantizer", "LlmActivationQuantizer", "LinearEntry", "ActivationEntry", "quantize_layer_weights",
           "quantize_layer_activations", "save_weight_checkpoint", "save_checkpoint", "load_checkpoint", "load_weight_checkpoint"]


@configclass
@dataclass
class LlmQuantizerConfig(SkipBasedConfig, ProgressiveQuantizerConfig):
    """app/llm/quant/quantizer/config.py:18-98: a (progressive) quantizer plus whether its range is static, the GPTQ
    kernel and the range calibration."""

    static: bool = False
    kernel_gptq: QuantGptqConfig | None = None
    calib_range: SkipBasedDynamicRangeCalibConfig | None = None

    def __post_init__(self) -> None:
        super().__post_init__()
        if self.quant_dtype is None:
            self.static, self.kernel_gptq, self.calib_range = False, None, None
        if self.static and self.calib_range is None:
            self.calib_range = SkipBasedDynamicRangeCalibConfig()

    @property
    def enabled_gptq(self) -> bool:
        return self.kernel_gptq is not None

    @property
    def enabled_calib_range(self) -> bool:
        return self.calib_range is not None

    @property
    def needs_calib_data(self) -> bool:
        return self.enabled_calib_range and (self.calib_range.needs_search or self.static)

    def generate_calib_dirname(self) -> str:
        name = ""
        if self.static:
            name += ".static"
        if self.enabled_gptq:
            name += ".gptq"
        if self.enabled_calib_range and (self.calib_range.needs_search or self.calib_range.ratio != 1):
            name += ".range"
        return name[1:] if name else ""


@configclass
@dataclass
class LlmWeightQuantizerConfig(LlmQuantizerConfig):
    """Weights are always quantized statically (config.py:101-124)."""

    static: bool = field(init=False, default=True)

    @property
    def needs_calib_data(self) -> bool:
        return self.enabled_calib_range and self.calib_range.needs_search


@configclass
@dataclass
class LlmActivationQuantizerConfig(LlmQuantizerConfig):
    """Activations never take the GPTQ kernel (config.py:127-160)."""

    kernel_gptq: None = field(init=False, default=None)


@dataclass
class LlmQuantizer(Quantizer):
    """quantizer.py:22-129: a Quantizer that takes its kernel and its range calibration from its config."""

    config: LlmQuantizerConfig = None
    tensor_type: TensorType = TensorType.Weights

    def __post_init__(self) -> None:
        self.kernel = self.config.kernel_gptq

    def calibrate_dynamic_range(self, modules: tp.Sequence[nn.Module], activations: TensorsCache,
                                weights: tp.Sequence[nn.Parameter] | None = None,
                                eval_inputs: TensorsCache | None = None, eval_module: nn.Module | None = None,
                                eval_kwargs: dict[str, tp.Any] | None = None, orig_weights=None, orig_activations=None,
                                orig_eval_inputs=None) -> tp.Sequence[DynamicRange] | None:
        if (not self.is_enabled() or self.config.calib_range is None
                or not self.config.calib_range.is_enabled_for(self.key)):
            self.dynamic_range = None
        else:
            self.dynamic_range = calibrate_dynamic_range(
                tensor_type=self.tensor_type, config=self.config.calib_range, static=self.config.static, quantizer=self,
                modules=modules, activations=activations, weights=weights, eval_inputs=eval_inputs,
                eval_module=eval_module, eval_kwargs=eval_kwargs, orig_weights=orig_weights,
                orig_activations=orig_activations, orig_eval_inputs=orig_eval_inputs)
        return self.dynamic_range


@dataclass
class LlmWeightQuantizer(LlmQuantizer):
    """quantizer.py:132-215."""

    config: LlmWeightQuantizerConfig = None
    channels_dim: None = field(init=False, default=None)
    tensor_type: TensorType = field(init=False, default=TensorType.Weights)

    def calibrate_dynamic_range(self, module: nn.Module, inputs: TensorsCache, weight: nn.Parameter | None = None,
                                eval_inputs: TensorsCache | None = None, eval_module: nn.Module | None = None,
                                eval_kwargs: dict[str, tp.Any] | None = None, orig_inputs: TensorsCache | None = None,
                                orig_eval_inputs: TensorsCache | None = None):
        return super().calibrate_dynamic_range(
            modules=[module], weights=[weight] if weight is not None else [module.weight], activations=inputs,
            eval_inputs=eval_inputs, eval_module=eval_module, eval_kwargs=eval_kwargs, orig_activations=orig_inputs,
            orig_eval_inputs=orig_eval_inputs)


@dataclass
class LlmActivationQuantizer(LlmQuantizer):
    """quantizer.py:218-258: inputs or outputs of a module, channels on ``channels_dim``."""

    config: LlmActivationQuantizerConfig = None
    tensor_type: TensorType = TensorType.Inputs

    def __post_init__(self) -> None:
        super().__post_init__()
        assert self.tensor_type != TensorType.Weights, "The tensor type cannot be weights."
        assert isinstance(self.channels_dim, int), "The channels dimension must be provided."


@dataclass
class LinearEntry:
    """One linear of a layer: ``key`` selects per-kind settings (qkv_proj, out_proj, up_proj, down_proj ...), ``name``
    is its path in the model (checkpoint keys derive from it).  ``eval_*``: the module whose outputs score range
    candidates and the name its inputs are cached under; default: the linear itself (the reference scores q_proj /
    k_proj on their whole attention block, weight.py:60-64)."""

    key: str
    name: str
    module: nn.Linear
    eval_name: str | None = None
    eval_module: nn.Module | None = None
    eval_kwargs: dict[str, tp.Any] | None = None


@torch.inference_mode()
def quantize_layer_weights(entries: tp.Sequence[LinearEntry], config: LlmWeightQuantizerConfig,
                           quantizer_state_dict: dict[str, tp.Any],
                           layer_cache: dict[str, IOTensorsCache] | None = None,
                           return_with_scale_state_dict: bool = False, develop_dtype: torch.dtype = torch.float32,
                           streams: int = 1, hessian_hook: tp.Callable[[int], tp.Any] | None = None,
                           world: int = 1, rank: int = 0,
                           prefetch: tp.Callable[[], tp.Any] | None = None) -> dict[str, torch.Tensor | float | None]:
    """app/llm/quant/weight.py:24-105.  Pass 1 calibrates the range of every linear that has no entry in
    ``quantizer_state_dict`` yet (and stores it there); pass 2 reloads each range, quantizes the weight with the
    configured kernel (GPTQ when set, fed by the linear's cached inputs) and writes the dequantized weight back into
    the module.  Returns the scale state dict when asked for.

    ``streams`` > 1 (no counterpart in the reference, same results): with the GPTQ kernel, the linears are grouped by
    the cached inputs they share (q/k/v, gate/up ...); the groups' Hessians are computed one after the other on the
    caller's stream, each with the whole chip, largest pipeline first, and every group's searches, inverse factor and
    sweeps start on a stream of their own as soon as that group's Hessian is there.  ``hessian_hook(channels)`` is
    called right before each of those Hessians is enqueued (a measurement hook); it may return a pair of
    ``torch.cuda.Event(enable_timing=True)``, which are then recorded around that Hessian's matrix-core launches.
    When no scale state dict is asked for, the weights of a group are quantized as one weight stacked along the output
    rows (``_quantize_stacked``: rows are independent given the Hessian, so the results are the same bit for bit, and
    one sweep stands where one per linear stood).

    ``prefetch`` (no counterpart in the reference; ``streams`` > 1 only, ignored with ``world`` > 1): called once, on
    the caller's stream, when this layer's own Hessians have been queued and before its pipelines are started.  Meant
    for ``prepare_layer_hessians(next layer's entries, next layer's cache)``: the next layer's Hessians -- matrix-core
    work that fills the chip and depends on nothing this layer produces (a layer's cached inputs are recorded before
    the layer before it is quantized) -- then run beside this layer's factorisations and sweeps, which are chains of
    small kernels, and the next call finds them prepared.

    ``world`` > 1 (no counterpart in the reference either): ONE layer split over the ``world`` processes of the default
    process group, one per GPU.  Every rank passes the same modules (full weights) and the cache of ITS share of the
    calibration tokens: the partial Hessians are all-reduced, every rank searches and quantizes rows
    ``row_shard(out_features, world, rank)`` of each linear (search per group, scales and the GPTQ sweep are
    independent per output row given the Hessian), and the dequantized slabs are all-gathered into every rank's
    modules.  ``quantizer_state_dict`` and the returned scale entries describe the rank's own rows."""
    from ..quantizer.kernel.gptq import clear_hessian_cache

    layer_cache = layer_cache or {}
    if world > 1:
        return _quantize_layer_weights_row_sharded(entries, config, quantizer_state_dict, layer_cache,
                                                   return_with_scale_state_dict, develop_dtype, streams, hessian_hook,
                                                   world, rank)
    try:
        if streams > 1 and layer_cache and config.enabled_gptq and len(entries) > 1:
            return _quantize_layer_weights_on_streams(entries, config, quantizer_state_dict, layer_cache,
                                                      return_with_scale_state_dict, develop_dtype, streams, hessian_hook,
                                                      prefetch)
        if prefetch is not None:
            prefetch()
        return _quantize_layer_weights(entries, config, quantizer_state_dict, layer_cache,
                                       return_with_scale_state_dict, develop_dtype)
    finally:
        # the GPTQ kernel memoises the Hessian (and its inverse factor) of the last activation cache together with a
        # reference to that cache; a layer's caches must not outlive the layer (what `prefetch` prepared for the next
        # layer lives until the end of that layer's call)
        clear_hessian_cache(keep_ahead=prefetch is not None)


def _groups_by_inputs(entries, layer_cache, device):
    """The linears grouped by the cached inputs they share, the costliest pipeline first."""
    from ..quantizer.kernel import gptq as gptq_mod

    groups: dict[tp.Any, list[LinearEntry]] = {}
    for e in entries:
        groups.setdefault(gptq_mod.inputs_key(layer_cache[e.name].inputs.front(), device), []).append(e)

    def cost(members):  # ~ C^3 for the inverse factor + R C^2 for the sweeps
        c = members[0].module.in_features
        return c * c * (c + sum(m.module.out_features for m in members))

    return sorted(groups.values(), key=cost, reverse=True), cost


def prepare_layer_hessians(entries: tp.Sequence[LinearEntry], layer_cache: dict[str, IOTensorsCache],
                           hessian_hook: tp.Callable[[int], tp.Any] | None = None) -> None:
    """Queue the Hessians of a layer that ``quantize_layer_weights(..., streams > 1)`` will be called for NEXT, on the
    current stream, in the order that call would compute them (largest pipeline first).  To be passed as the
    ``prefetch`` of the call for the layer before it; they are kept until the end of the next call."""
    from ..quantizer.kernel import gptq as gptq_mod

    device = entries[0].module.weight.device
    order, _ = _groups_by_inputs(entries, layer_cache, device)
    # beside the running layer's chains of single-workgroup kernels: an eighth of the compute units is left alone
    width = torch.cuda.get_device_properties(device).multi_processor_count * 7 // 8
    for members in order:
        events = hessian_hook(members[0].module.in_features) if hessian_hook is not None else None
        gptq_mod.prepare_hessians([layer_cache[members[0].name].inputs.front()], device, round_width=width,
                                  profile_events=events, ahead=True)


def _row_multiple(config: LlmWeightQuantizerConfig) -> int:
    """Rows that must stay together in a slab: the largest row extent of any quantization group (a group shape of
    (1, g) cuts nowhere; whole-tensor or whole-column groups cannot be row-sharded)."""
    multiple = 1
    for shape in config.group_shapes:
        if shape[0] <= 0:
            raise ValueError("groups that span all output rows cannot be sharded by rows")
        multiple = max(multiple, int(shape[0]))
    return multiple


def _quantize_layer_weights_row_sharded(entries, config, quantizer_state_dict, layer_cache,
                                        return_with_scale_state_dict, develop_dtype, streams, hessian_hook, world, rank):
    from ..parallel import all_gather_slabs, row_shard
    from ..quantizer.kernel import gptq as gptq_mod

    multiple = _row_multiple(config)
    device = entries[0].module.weight.device
    full = [e.module.weight.data for e in entries]
    bounds = [[row_shard(w.shape[0], world, r, multiple) for r in range(world)] for w in full]
    try:
        if layer_cache and config.enabled_gptq:
            # token-sharded Hessians, the same order on every rank (collectives must line up): largest first
            seen, caches = set(), []
            for e in sorted(entries, key=lambda e: -e.module.in_features):
                cache = layer_cache[e.name].inputs.front()
                key = gptq_mod.inputs_key(cache, device)
                if key not in seen:
                    seen.add(key)
                    caches.append((e.module.in_features, cache))
            for channels, cache in caches:
                events = hessian_hook(channels) if hessian_hook is not None else None
                gptq_mod.prepare_hessians([cache], device, world=world, profile_events=events)
        for e, w, b in zip(entries, full, bounds):
            begin, end = b[rank]
            e.module.weight.data = w[begin:end]  # a contiguous view of whole rows
        if streams > 1 and layer_cache and config.enabled_gptq and len(entries) > 1:
            out = _quantize_layer_weights_on_streams(entries, config, quantizer_state_dict, layer_cache,
                                                     return_with_scale_state_dict, develop_dtype, streams, None)
        else:
            out = _quantize_layer_weights(entries, config, quantizer_state_dict, layer_cache,
                                          return_with_scale_state_dict, develop_dtype)
        for e, w, b in zip(entries, full, bounds):
            e.module.weight.data = all_gather_slabs(e.module.weight.data, [hi - lo for lo, hi in b], world, rank)
        return out
    except BaseException:
        for e, w in zip(entries, full):
            e.module.weight.data = w
        raise
    finally:
        gptq_mod.clear_hessian_cache()


def _quantize_layer_weights_on_streams(entries, config, quantizer_state_dict, layer_cache, return_with_scale_state_dict,
                                       develop_dtype, streams, hessian_hook,
                                       prefetch=None) -> dict[str, torch.Tensor | float | None]:
    from ..parallel import assign_by_cost, run_on_streams
    from ..quantizer.kernel import gptq as gptq_mod

    device = entries[0].module.weight.device
    order, cost = _groups_by_inputs(entries, layer_cache, device)
    start = torch.cuda.Event()
    start.record()
    ready = []
    ahead = []  # the group's Hessian was prepared during the previous layer's call
    # The first (largest) Hessian has the chip to itself.  The later ones run beside the first group's factorisation,
    # a chain of single-workgroup kernels that each need a free compute unit to start on, so they leave an eighth of
    # the compute units alone (measured: 120-122 -> 118-119 ms per Llama-3-8B layer; 3/4 and 15/16 gain less).
    width = torch.cuda.get_device_properties(device).multi_processor_count * 7 // 8
    for n, members in enumerate(order):
        inputs = layer_cache[members[0].name].inputs.front()
        ahead.append(gptq_mod.is_prepared(inputs, device))
        events = hessian_hook(members[0].module.in_features) if hessian_hook is not None and not ahead[-1] else None
        hessian = gptq_mod.prepare_hessians([inputs], device, round_width=width if n > 0 else 0, profile_events=events)[0]
        ready.append(getattr(hessian, "_dcq_ready", None))
        if ready[-1] is None:
            ready[-1] = torch.cuda.Event()
            ready[-1].record()
    if prefetch is not None:
        prefetch()
    # longest pipeline first, the rest dealt to the least loaded stream (each list ascending = order of readiness)
    buckets = [b for b in assign_by_cost([float(cost(members)) for members in order], min(streams, len(order))) if b]

    pending: list = []  # host copies of the searched ranges and zeros, taken when every pipeline has been enqueued

    def run_bucket(indices):
        parts = []
        for i in indices:
            wait = lambda i=i: torch.cuda.current_stream().wait_event(ready[i])  # noqa: E731
            # The searches need the weights only.  Those of the first group go out before its Hessian is waited for:
            # they are bound by the vector pipe and run beside the transposed copy of the wide activations (bound by
            # HBM) that precedes the first SYRK, instead of between that SYRK and the group's factorisation (-0.65 ms
            # per Llama-3-8B layer over six A/B pairs).  For the later groups the same move only takes compute units
            # from a SYRK.
            early = i == 0 and not ahead[0]
            if not early:
                wait()
            parts.append(_quantize_layer_weights(order[i], config, quantizer_state_dict, layer_cache,
                                                 return_with_scale_state_dict, develop_dtype, host_copies=pending,
                                                 stack=True, before_quantizing=wait if early else None))
        return parts

    # The first bucket holds the longest pipeline (down_proj's: its Hessian is the first and by far the largest, its
    # factorisation and sweep the longest chains), and since the groups are quantized as stacked weights the others
    # finish a few milliseconds before it: its stream gets the higher priority (A/B, six pairs on two boxes: 100.9-103.4
    # -> 99.3-102.1 ms per Llama-3-8B layer, -1.5 ms on average; before the stacking the same switch only moved time
    # from one path to the other).
    prio = [-1] + [0] * (len(buckets) - 1)
    results = run_on_streams([lambda b=b: run_bucket(b) for b in buckets], device=device,
                             # (a Hessian prepared ahead was ready before this call's inputs may have been: `start`)
                             after=[start if b[0] == 0 or ahead[b[0]] else ready[b[0]] for b in buckets],
                             priorities=prio)
    for copy in pending:
        copy()
    merged: dict[str, torch.Tensor | float | None] = {}
    for bucket in results:
        for part in bucket:
            merged.update(part)
    # keys in the order the serial stage produces them
    names = [f"{e.name}.weight." for e in entries]
    return dict(sorted(merged.items(), key=lambda kv: next((i for i, n in enumerate(names) if kv[0].startswith(n)),
                                                          len(names))))


def _stack_ranges(ranges: list, rows: list[int]):
    """The dynamic ranges of several weights as the range of the weights stacked along their rows, or None when they
    do not stack: every weight must bring the same kind of range (nothing, a number per tensor, or per-group tensors
    whose leading extent is the weight's number of row groups)."""
    if all(r is None for r in ranges):
        return None
    if any(r is None for r in ranges):
        return False
    ranges = [(r,) if isinstance(r, DynamicRange) else tuple(r) for r in ranges]
    if len({len(r) for r in ranges}) != 1:
        return False
    stacked = []
    for step in zip(*ranges):
        if all(r is None for r in step):
            stacked.append(None)
            continue
        if any(r is None for r in step):
            return False
        parts = {}
        for field_name in ("min", "max", "ratio"):
            values = [getattr(r, field_name) for r in step]
            if all(v is None for v in values):
                parts[field_name] = None
                continue
            if any(v is None for v in values):
                return False
            tensors = [v for v in values if isinstance(v, torch.Tensor)]
            if not tensors:
                if len(set(float(v) for v in values)) == 1:
                    parts[field_name] = values[0]  # the same number everywhere
                    continue
                return False  # different numbers per tensor: no per-group shape to spread them over
            template = tensors[0]
            if template.ndim < 2 or any(isinstance(v, torch.Tensor) and (v.ndim != template.ndim or v.shape[1:] != template.shape[1:])
                                        for v in values):
                return False
            pieces = []
            for v, n in zip(values, rows):
                if not isinstance(v, torch.Tensor):
                    v = torch.full((n, *template.shape[1:]), float(v), dtype=template.dtype, device=template.device)
                if v.shape[0] != n:  # row groups taller than one row, or a per-tensor value: not stacked here
                    return False
                pieces.append(v)
            parts[field_name] = torch.cat(pieces, dim=0)
        stacked.append(DynamicRange(**parts))
    return tuple(stacked)


def _quantize_stacked(entries, quantizers, layer_cache) -> bool:
    """Weights fed by the same activations, quantized by the same kind of quantizer, as ONE weight stacked along the
    output rows (no counterpart in the reference; same results: every row of a weight is quantized on its own given the
    Hessian, per row group for the scales).  One sweep -- a serial chain of small kernels whose length hardly depends on
    the number of rows -- then stands where one per weight stood.  False: the weights do not stack, nothing was done."""
    weights = [e.module.weight.data for e in entries]
    first = weights[0]
    if any(w.ndim != 2 or w.shape[1] != first.shape[1] or w.dtype != first.dtype or w.device != first.device
           for w in weights):
        return False
    q0 = quantizers[0]
    if any(q.config is not q0.config or q.develop_dtype != q0.develop_dtype or q.channels_dim != q0.channels_dim
           or q.range_bound != q0.range_bound or q.quant_range != q0.quant_range or q.scale is not None
           or q.zero is not None for q in quantizers):
        return False
    if any(shape[0] != 1 for shape in q0.config.group_shapes):  # groups must not reach over rows
        return False
    kernel = q0.kernel
    if kernel is None or any(hasattr(kernel, "is_enabled_for") and not kernel.is_enabled_for(q.key) for q in quantizers):
        return False
    rows = [w.shape[0] for w in weights]
    ranges = _stack_ranges([q.dynamic_range for q in quantizers], rows)
    if ranges is False:
        return False
    result = q0.quantize(torch.cat(weights, dim=0), inputs=layer_cache[entries[0].name].inputs.front(),
                         dynamic_range=ranges, return_with_dequant=True, return_with_quant=False)
    at = 0
    for e, n in zip(entries, rows):
        e.module.weight.data = result.data[at:at + n]
        at += n
    return True


def _quantize_layer_weights(entries, config, quantizer_state_dict, layer_cache, return_with_scale_state_dict,
                            develop_dtype, host_copies: list | None = None, stack: bool = False,
                            before_quantizing=None) -> dict[str, torch.Tensor | float | None]:
    # Ranges searched here stay on the device for the second pass; their host copies for ``quantizer_state_dict``
    # (the contents of wgts.pt) are taken after the last kernel of the stage is enqueued, so that no search is waited
    # for before the work behind it is queued.  ``host_copies``: a list that receives those pending copies as
    # callables instead (the caller runs them once all of its pipelines are done).
    searched: dict[str, LlmWeightQuantizer] = {}
    for e in entries:
        assert isinstance(e.module, nn.Linear)
        quantizer = LlmWeightQuantizer(config, develop_dtype=develop_dtype, key=e.key)
        if quantizer.is_enabled() and e.name not in quantizer_state_dict:
            eval_name = e.eval_name or e.name
            quantizer.calibrate_dynamic_range(
                module=e.module, inputs=layer_cache[e.name].inputs if layer_cache else None,
                eval_inputs=layer_cache[eval_name].inputs if layer_cache else None,
                # without cached activations only a search on the weight itself can run: no evaluation module then
                eval_module=(e.eval_module or e.module) if layer_cache else None, eval_kwargs=e.eval_kwargs)
            searched[e.name] = quantizer
    if before_quantizing is not None:  # (the searches above need the weights only; what follows may need more)
        before_quantizing()
    scale_state_dict: dict[str, torch.Tensor | float | None] = {}
    zeros: list[tuple[str, torch.Tensor]] = []
    stacked = False
    if stack and len(entries) > 1 and layer_cache and not return_with_scale_state_dict:
        # ``stack``: the caller vouches that the entries share their cached inputs
        quantizers = []
        for e in entries:
            quantizer = searched.get(e.name) or LlmWeightQuantizer(config, develop_dtype=develop_dtype, key=e.key)
            if e.name not in searched and quantizer.is_enabled():
                quantizer.load_state_dict(quantizer_state_dict[e.name], device=e.module.weight.device)
            quantizers.append(quantizer)
        if all(q.is_enabled() for q in quantizers):
            stacked = _quantize_stacked(entries, quantizers, layer_cache)
    for e in entries if not stacked else ():
        quantizer = searched.get(e.name) or LlmWeightQuantizer(config, develop_dtype=develop_dtype, key=e.key)
        if not quantizer.is_enabled():
            continue
        param_name = f"{e.name}.weight"
        if e.name not in searched:
            quantizer.load_state_dict(quantizer_state_dict[e.name], device=e.module.weight.device)
        result = quantizer.quantize(e.module.weight.data,
                                    inputs=layer_cache[e.name].inputs.front() if layer_cache else None,
                                    return_with_dequant=True, return_with_quant=return_with_scale_state_dict)
        e.module.weight.data = result.data
        if return_with_scale_state_dict:
            scale_state_dict.update(result.scale.state_dict(f"{param_name}.scale"))
            zero_name = "scaled_zero" if config.zero_domain is ZeroPointDomain.PostScale else "zero"
            scale_state_dict[f"{param_name}.{zero_name}"] = result.zero
            if isinstance(result.zero, torch.Tensor):
                zeros.append((f"{param_name}.{zero_name}", result.zero))
    def copy_to_host() -> None:
        # every copy queued, then one wait for all of them (page-locked destinations)
        for name, quantizer in searched.items():
            quantizer_state_dict[name] = quantizer.state_dict(non_blocking=True)
        for key, zero in zeros:
            scale_state_dict[key] = (torch.empty(zero.shape, dtype=zero.dtype, pin_memory=True).copy_(
                zero, non_blocking=True) if zero.is_cuda else zero.clone())
        if torch.cuda.is_available() and (searched or zeros):
            torch.cuda.current_stream().synchronize()

    if host_copies is None:
        copy_to_host()
    else:
        host_copies.append(copy_to_host)
    return scale_state_dict


@dataclass
class ActivationEntry:
    """One activation tensor of a layer to quantize: the inputs shared by ``modules`` (``TensorType.Inputs``; the
    quantizer is hooked in front of each of them) or the outputs of ``modules`` (``TensorType.Outputs``; hooked behind
    them).  ``name`` is the module name the activations are cached under (for shared inputs: the module whose input
    cache is used, the reference takes v_proj / the first up projection).  ``eval_*`` as in :class:`LinearEntry`;
    ``orig_weights``: (parameter, original tensor) pairs put in place while the outputs to match are computed."""

    key: str
    tensor_type: TensorType
    modules: tp.Sequence[nn.Module]
    name: str
    eval_module: nn.Module | None = None
    eval_name: str | None = None
    eval_kwargs: dict[str, tp.Any] | None = None
    orig_weights: list[tuple[nn.Parameter, torch.Tensor]] | None = None


@torch.inference_mode()
def quantize_layer_activations(entries: tp.Sequence[ActivationEntry], ipts_config: LlmActivationQuantizerConfig | None,
                               opts_config: LlmActivationQuantizerConfig | None,
                               quantizer_state_dict: dict[str, tp.Any],
                               layer_cache: dict[str, IOTensorsCache] | None = None,
                               develop_dtype: torch.dtype = torch.float32, register_hooks: bool = True
                               ) -> dict[str, tuple[LlmActivationQuantizer, list]]:
    """app/llm/quant/activation.py:129-192.  For every entry whose quantizer is enabled: calibrate its range on the
    cached activations unless ``quantizer_state_dict`` (``acts.pt``) already has it under ``"{name}.input"`` /
    ``"{name}.output"``, else reload it; then hook the quantizer onto the entry's modules so that later forwards see
    fake-quantized activations.  Returns ``{cache key: (quantizer, hooks)}``."""
    layer_cache = layer_cache or {}
    out: dict[str, tuple[LlmActivationQuantizer, list]] = {}
    for e in entries:
        inputs = e.tensor_type == TensorType.Inputs
        cache_key = f"{e.name}.input" if inputs else f"{e.name}.output"
        config = ipts_config if inputs else opts_config
        if config is None:
            continue
        quantizer = LlmActivationQuantizer(config, channels_dim=-1, develop_dtype=develop_dtype, key=e.key,
                                           tensor_type=e.tensor_type)
        if not quantizer.is_enabled():
            continue
        if cache_key not in quantizer_state_dict:
            io = layer_cache.get(e.name, IOTensorsCache())
            eval_name = e.eval_name or e.name
            quantizer.calibrate_dynamic_range(
                modules=list(e.modules), activations=io.inputs if inputs else io.outputs,
                eval_module=e.eval_module, eval_inputs=layer_cache[eval_name].inputs if layer_cache else None,
                eval_kwargs=e.eval_kwargs, orig_weights=e.orig_weights)
            quantizer_state_dict[cache_key] = quantizer.state_dict()
        else:
            device = next((p.device for m in e.modules for p in m.parameters()), "cuda")
            quantizer.load_state_dict(quantizer_state_dict[cache_key], device=device)
        hooks = []
        if register_hooks:
            hooks = [quantizer.as_hook(is_output=not inputs).register(m) for m in e.modules]
        out[cache_key] = (quantizer, hooks)
    return out


CHECKPOINT_FILES = ("rotation.pt", "reorder.pt", "smooth.pt", "branch.pt", "wgts.pt", "acts.pt", "scale.pt", "model.pt")


def save_checkpoint(dirpath: str, *, rotation: torch.Tensor | None = None,
                    reorder: dict[str, torch.Tensor] | None = None, smooth: dict[str, torch.Tensor] | None = None,
                    wgts: dict[str, tp.Any] | None = None, acts: dict[str, tp.Any] | None = None,
                    scale: dict[str, tp.Any] | None = None, model: dict[str, torch.Tensor] | None = None,
                    branch: dict[str, dict[str, torch.Tensor]] | None = None) -> dict[str, str]:
    """Write the stage results a run of the reference leaves in its cache / save directory (app/llm/ptq.py:74-78,
    99-103, 121-306; app/diffusion/ptq.py:205-222 for ``branch.pt``), one ``torch.save`` per file: ``rotation.pt`` the
    rotation matrix; ``reorder.pt`` channel indexes by module name; ``smooth.pt`` smoothing scales by module name;
    ``branch.pt`` the low-rank branches' state dicts (``a.weight`` [rank, in], ``b.weight`` [out, rank]) by the name
    of the first module that shares the branch; ``wgts.pt`` / ``acts.pt`` the weight / activation quantizers' state
    dicts by module name; ``scale.pt`` the flattened ``{param}.scale.{level}`` / ``zero`` / ``scaled_zero`` tensors;
    ``model.pt`` the (fake-quantized) model's state dict.  Tensors are written from host memory, so a file never names
    a GPU of the machine that wrote it.  Returns the paths written."""
    os.makedirs(dirpath, exist_ok=True)
    parts = dict(rotation=rotation, reorder=reorder, smooth=smooth, branch=branch, wgts=wgts, acts=acts, scale=scale,
                 model=model)
    written = {}
    for name, obj in parts.items():
        if obj is not None:
            path = os.path.join(dirpath, f"{name}.pt")
            torch.save(_to_host(obj), path)
            written[f"{name}.pt"] = path
    return written


def _to_host(obj):
    """The same nest of dicts / lists / tuples with every tensor in host memory."""
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return type(obj)((k, _to_host(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v) for v in obj)
    return obj


def load_checkpoint(dirpath: str, map_location: str | torch.device | None = "cpu") -> dict[str, tp.Any]:
    """Whatever of the files of ``save_checkpoint`` exists in ``dirpath``, by file name.  The files hold tensors,
    numbers, None and plain containers only, so they are read with ``weights_only=True``; the rotation matrix comes
    back in float64, as app/llm/ptq.py:121 converts it.  ``map_location`` as in ``torch.load`` (host memory unless
    told otherwise, so files written on another machine load anywhere)."""
    out = {}
    for fname in CHECKPOINT_FILES:
        path = os.path.join(dirpath, fname)
        if os.path.exists(path):
            out[fname] = torch.load(path, weights_only=True, map_location=map_location)
    if "rotation.pt" in out:
        out["rotation.pt"] = out["rotation.pt"].to(dtype=torch.float64)
    return out


def save_weight_checkpoint(dirpath: str, *, quantizer_state_dict: dict[str, tp.Any] | None = None,
                           scale_state_dict: dict[str, tp.Any] | None = None,
                           model_state_dict: dict[str, torch.Tensor] | None = None) -> dict[str, str]:
    """``wgts.pt`` / ``scale.pt`` / ``model.pt`` of the weight stage (app/llm/ptq.py:251-266)."""
    return save_checkpoint(dirpath, wgts=quantizer_state_dict, scale=scale_state_dict, model=model_state_dict)


def load_weight_checkpoint(dirpath: str) -> dict[str, tp.Any]:
    """Whatever of ``wgts.pt`` / ``scale.pt`` / ``model.pt`` exists in ``dirpath``."""
    return {k: v for k, v in load_checkpoint(dirpath).items() if k in ("wgts.pt", "scale.pt", "model.pt")}
