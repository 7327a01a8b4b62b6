"""Smooth quantization (reference: deepcompressor/calib/smooth.py:34-1109).

Same call surface as the reference: ``smooth_linear_modules`` (weight-centric, x-w) and ``smooth_attention``
(outputs-centric, query-key) with a given scale or a ``SmoothLinearCalibrator`` / ``SmoothAttentionCalibrator``
search (Manual / GridSearch over (alpha, beta) x span pairs), scored on the outputs of an evaluation module (Layer
granularity) or on the x-w products of the linears themselves (ProductsError: Layer, or one winner per group of input
channels).  What runs where:

* per-channel spans of activations and weights        -> ``dcq_channel_metric``   (one pass each, HBM-bound)
* ``W * s`` / ``W / s`` / ``x / s`` / ``x * s`` (+ cast) -> ``dcq_scale_channels``   (one pass, fused cast)
* fake quantization of the scaled tensors             -> the ``Quantizer`` kernels
* the evaluation module's forward                     -> torch on the GPU (the module belongs to the caller)
* products of the processed x and w per channel group -> one batched GEMM, errors by ``dcq_product_error``

Not implemented (raise): TensorError scoring of smoothing candidates, ProductsError on convolution weights.
"""

from __future__ import annotations

import ctypes as C
import gc
import typing as tp
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import engine
from .._lib import check, lib, storage_of, stream_of
from ..data.cache import TensorsCache
from ..data.common import TensorType
from ..data.utils.shape import infer_view_shape
from ..quantizer.processor import ProcessHook, Quantizer
from .config.search import SearchBasedCalibGranularity, SearchBasedCalibObjective
from .config.smooth import SmoothCalibConfig, SmoothSpanMode
from .metric import ChannelMetric

__all__ = ["ActivationSmoother", "get_smooth_span", "get_smooth_scale", "scale_channels", "SmoothCalibrator",
           "SmoothLinearCalibrator", "SmoothAttentionCalibrator", "smooth_upscale_param", "smooth_downscale_param",
           "convert_smooth_upscale_to_downscale", "smooth_linear_modules", "smooth_attention"]


def scale_channels(tensor: torch.Tensor, scale: torch.Tensor, channels_dim: int, *, divide: bool,
                   out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """``(tensor.float() * scale).to(out_dtype)`` (or ``/``) with ``scale`` along ``channels_dim``; channels past
    ``scale.numel()`` are only cast.  One fused pass on the GPU."""
    out_dtype = out_dtype or tensor.dtype
    src = tensor.contiguous()
    dim = channels_dim % src.ndim
    outer = int(torch.Size(src.shape[:dim]).numel())
    channels = src.shape[dim]
    inner = int(torch.Size(src.shape[dim + 1:]).numel())
    scale = scale.to(device=src.device, dtype=torch.float32).contiguous().view(-1)
    assert scale.numel() <= channels, "more scale entries than channels"
    out = torch.empty(src.shape, dtype=out_dtype, device=src.device)
    if src.numel():
        check(lib().dcq_scale_channels(C.c_void_p(src.data_ptr()), storage_of(src.dtype), outer, channels, inner,
                                       C.c_void_p(scale.data_ptr()), scale.numel(), int(divide),
                                       C.c_void_p(out.data_ptr()), storage_of(out_dtype), stream_of(src.device)),
              "dcq_scale_channels")
    return out


@dataclass
class ActivationSmoother:
    """Divides (or multiplies) a tensor by the smoothing scale along ``channels_dim`` (smooth.py:34-78)."""

    smooth_scale: torch.Tensor
    channels_dim: int
    upscale: bool = False
    develop_dtype: torch.dtype | None = None
    input_packager: tp.Any = None
    output_packager: tp.Any = None

    def is_enabled(self) -> bool:
        return self.smooth_scale is not None

    def get_input_packager(self):
        return self.input_packager

    def get_output_packager(self):
        return self.output_packager

    def process(self, tensor: torch.Tensor) -> torch.Tensor:
        if self.develop_dtype is None:
            self.develop_dtype = tensor.dtype
        if self.develop_dtype != torch.float32:
            scale = self.smooth_scale.to(device=tensor.device, dtype=self.develop_dtype)
            shape = [1] * tensor.ndim
            shape[self.channels_dim] = -1
            t = tensor.to(dtype=self.develop_dtype)
            return (t.mul(scale.view(shape)) if self.upscale else t.div(scale.view(shape))).to(dtype=tensor.dtype)
        return scale_channels(tensor, self.smooth_scale, self.channels_dim, divide=not self.upscale)

    def as_hook(self, func=None, *, is_output: bool = False) -> ProcessHook:
        return ProcessHook(self, func, is_output=is_output)


@torch.inference_mode()
def get_smooth_span(tensors: tp.Sequence[torch.Tensor], /, *, group_shape: tp.Sequence[int], span_mode: SmoothSpanMode,
                    device: torch.device | str | None = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Per-channel span of tensors whose dimension 1 is the channel (smooth.py:81-113)."""
    fn = {SmoothSpanMode.AbsMax: ChannelMetric.abs_max, SmoothSpanMode.RootMeanSquare: ChannelMetric.root_mean_square}
    return fn[span_mode](tensors, tensors[0].shape[1], group_shape, device=device, dtype=dtype)


@torch.inference_mode()
def get_smooth_scale(*, alpha_base: torch.Tensor, beta_base: torch.Tensor, alpha: float, beta: float) -> torch.Tensor:
    """alpha_base^alpha / beta_base^beta; zeros -> 1; any NaN / Inf -> all ones (smooth.py:116-146).
    The powers are evaluated in float64 and rounded to the span dtype (within one unit in the last place of the
    reference's float32 ``pow``)."""
    assert 0 <= alpha <= 1 and 0 <= beta <= 1, "The smooth factors should be in [0, 1]."
    power = lambda base, e: base.to(torch.float64).pow(e).to(base.dtype)  # noqa: E731
    if alpha > 0:
        smooth_scale = power(alpha_base, alpha)
        if beta > 0:
            smooth_scale = smooth_scale.div_(power(beta_base, beta))
    else:
        smooth_scale = power(beta_base, -beta)
    # (selected on the device: host-side tests would drain the stream for every candidate of a search)
    smooth_scale = smooth_scale.masked_fill_(smooth_scale == 0, 1)
    return torch.where(torch.isfinite(smooth_scale).all(), smooth_scale, torch.ones_like(smooth_scale))


def _head_reduce(span: torch.Tensor, unique_heads: int, repeats: int, use_max: bool) -> torch.Tensor:
    """Grouped-query attention: heads that share a key/value head share one span (smooth.py:315-323, 340-343)."""
    span = span.view(unique_heads, repeats, -1)
    span = span.amax(dim=1, keepdim=True) if use_max else span.mean(dim=1, keepdim=True)
    return span.expand(unique_heads, repeats, -1).reshape(-1)


class SmoothCalibrator:
    """Search over smoothing scales for x-w products (``tensor_type`` Weights) or y-x products such as query-key
    (``tensor_type`` Outputs).  Ask / tell structure, candidate order and tie rule follow the reference
    (smooth.py:149-714, search.py:63-1078)."""

    def __init__(self, tensor_type: TensorType, config: SmoothCalibConfig, w_quantizer: Quantizer | None,
                 x_quantizer: Quantizer | None, y_quantizer: Quantizer | None, num_heads: int = 1,
                 num_head_repeats: int = 1, with_rope: bool = False, develop_dtype: torch.dtype = torch.float32) -> None:
        assert tensor_type in (TensorType.Weights, TensorType.Outputs)
        self.tensor_type, self.config, self.develop_dtype = tensor_type, config, develop_dtype
        self.objective, self.granularity = config.objective, config.granularity
        self.w_quantizer, self.x_quantizer, self.y_quantizer = w_quantizer, x_quantizer, y_quantizer
        enabled = lambda q: q is not None and q.is_enabled()  # noqa: E731
        self.needs_w_quant, self.needs_x_quant, self.needs_y_quant = map(enabled, (w_quantizer, x_quantizer, y_quantizer))
        self.needs_x_quant_for_wgts = config.allow_a_quant and self.needs_x_quant
        self.needs_w_quant_for_wgts = config.allow_b_quant and self.needs_w_quant
        self.needs_x_quant_for_opts = config.allow_b_quant and self.needs_x_quant  # smooth.py:242-250
        self.needs_y_quant_for_opts = config.allow_a_quant and self.needs_y_quant
        self.num_heads, self.num_head_repeats = num_heads, num_head_repeats
        self.with_rope = tensor_type != TensorType.Weights and with_rope

        def largest(q, needed):
            return list(q.config.largest_group_shape) if needed else [1, None, -1]

        w_gs, x_gs, y_gs = largest(w_quantizer, self.needs_w_quant), largest(x_quantizer, self.needs_x_quant), \
            largest(y_quantizer, self.needs_y_quant)
        w_gs[1] = x_gs[1] if w_gs[1] is None else w_gs[1]
        x_gs[1] = (w_gs[1] if tensor_type == TensorType.Weights else y_gs[1]) if x_gs[1] is None else x_gs[1]
        y_gs[1] = x_gs[1] if y_gs[1] is None else y_gs[1]
        self.w_group_shape, self.x_group_shape, self.y_group_shape = w_gs, x_gs, y_gs
        self.alpha_beta_pairs = config.get_alpha_beta_pairs()
        self.num_iters = 1
        # calibration tokens sharded over processes: ``reduce_max(t)`` / ``reduce_sum(t)`` return the maximum / sum of
        # a small tensor over the shards.  Abs-max spans of activations take the maximum (exact: the scales equal the
        # unsharded ones), every candidate's error the sum, so all shards pick the same candidate.
        self.reduce_max = self.reduce_sum = None

    # ---- candidates
    @property
    def population_size(self) -> int:
        return len(self.alpha_beta_pairs) * len(self.config.spans)

    def is_done(self) -> bool:
        return self.iter >= self.num_iters

    def reset(self, *, x_wgts: list[torch.Tensor], x_acts: TensorsCache, y_wgts: list[torch.Tensor] | None = None,
              y_acts: TensorsCache | None = None, **kwargs) -> None:
        """smooth.py:272-386."""
        self.iter = self.candidate_id = 0
        self._state_dict: list[tuple[nn.Parameter, torch.Tensor]] = []
        self._hooks: list = []
        wgts_centric = self.tensor_type == TensorType.Weights
        self.num_in_channels = x_wgts[0].shape[1] if wgts_centric else x_wgts[0].shape[0]
        device = x_wgts[0].device
        grouped = self.num_heads > 1 and self.num_head_repeats > 1
        self.num_unique_heads = self.num_heads // self.num_head_repeats if grouped else 0
        assert x_acts.num_tensors == 1, f"Only one input is allowed, got {x_acts.num_tensors}"
        x_tensors = x_acts.front().get_standardized_data(reshape=False)
        assert all(x.shape[1] == self.num_in_channels for x in x_tensors)

        def rope_reduce(span, use_max):
            # rotary embedding pairs channel i with i + head_dim / 2: both halves of a head share one span
            span = span.view(self.num_heads, 2, -1)
            span = span.amax(dim=1, keepdim=True) if use_max else span.mean(dim=1, keepdim=True)
            return span.expand(self.num_heads, 2, -1).reshape(-1)

        def spans_of(tensors, group_shape, modes, *, heads: str, rope: bool, sharded: bool = True):
            out = {}
            for mode in modes:
                span = get_smooth_span(tensors, group_shape=group_shape, span_mode=mode, device=device,
                                       dtype=self.develop_dtype)
                if sharded and self.reduce_max is not None:
                    if "Max" not in mode.name:
                        raise NotImplementedError("token-sharded spans are implemented for the AbsMax span only")
                    span = self.reduce_max(span)
                if self.num_unique_heads > 0:
                    if heads == "reduce":
                        span = _head_reduce(span, self.num_unique_heads, self.num_head_repeats, "Max" in mode.name)
                    else:  # one span per key/value head, repeated for the query heads that share it
                        span = span.view(self.num_unique_heads, 1, -1).expand(
                            self.num_unique_heads, self.num_head_repeats, -1).reshape(-1)
                if rope:
                    span = rope_reduce(span, "Max" in mode.name)
                out[mode] = span
            return out

        rope = not wgts_centric and self.with_rope
        x_spans = spans_of(x_tensors, self.x_group_shape, self.config.a_spans if wgts_centric else self.config.b_spans,
                           heads="reduce", rope=rope)
        if wgts_centric:
            assert all(w.shape[1] == self.num_in_channels for w in x_wgts)
            w_spans = spans_of([w.data for w in x_wgts], self.w_group_shape, self.config.b_spans, heads="reduce",
                               rope=False, sharded=False)  # the weights are the same on every shard
            self.span_pairs = [(x_spans[a], w_spans[b]) for a, b in self.config.spans]
        else:
            assert y_acts is not None and y_acts.num_tensors == 1, "Only one output source is allowed"
            num_out_channels = (self.num_in_channels // self.num_head_repeats if self.num_unique_heads > 0
                                else self.num_in_channels)
            assert all(w.shape[0] == self.num_in_channels for w in x_wgts)
            assert all(w.shape[0] == num_out_channels for w in y_wgts)
            y_tensors = y_acts.front().get_standardized_data(reshape=False)
            assert all(y.shape[1] == num_out_channels for y in y_tensors)
            y_spans = spans_of(y_tensors, self.x_group_shape, self.config.a_spans, heads="expand", rope=rope)
            self.span_pairs = [(y_spans[a], x_spans[b]) for a, b in self.config.spans]
        self._scales: torch.Tensor | None = None  # the candidates of these spans (_candidate_table)
        self.best_error: list[torch.Tensor] | None = None
        self.best_scale: torch.Tensor | None = None
        self.error_history: list[list[torch.Tensor]] = []  # every candidate's errors, in candidate order

    def get_best(self) -> torch.Tensor:
        return self.best_scale

    def _candidate_table(self) -> torch.Tensor:
        """Every candidate scale of the grid, [candidates][channels] in the develop dtype: row i is
        ``get_smooth_scale`` of the i-th (alpha, beta) pair inside each pair of spans, all rows from a handful of
        batched operations instead of one chain of small launches per candidate (float64 powers rounded to the span
        dtype, as ``get_smooth_scale`` evaluates them)."""
        if self._scales is not None:
            return self._scales
        device = self.span_pairs[0][0].device
        alphas = torch.tensor([a for a, _ in self.alpha_beta_pairs], dtype=torch.float64, device=device).view(-1, 1)
        betas = torch.tensor([b for _, b in self.alpha_beta_pairs], dtype=torch.float64, device=device).view(-1, 1)
        assert all(0 <= a <= 1 and 0 <= b <= 1 for a, b in self.alpha_beta_pairs), \
            "The smooth factors should be in [0, 1]."
        tables = []
        for a_span, b_span in self.span_pairs:
            dtype = a_span.dtype
            a64, b64 = a_span.to(torch.float64).view(1, -1), b_span.to(torch.float64).view(1, -1)
            pa, pb, nb = a64.pow(alphas).to(dtype), b64.pow(betas).to(dtype), b64.pow(-betas).to(dtype)
            scale = torch.where(alphas > 0, torch.where(betas > 0, pa / pb, pa), nb)
            scale = scale.masked_fill_(scale == 0, 1)
            finite = torch.isfinite(scale).all(dim=1, keepdim=True)
            scale = torch.where(finite & ((alphas != 0) | (betas != 0)), scale, torch.ones_like(scale))
            tables.append(scale.to(self.develop_dtype))
        self._scales = torch.cat(tables)
        return self._scales

    def _candidate_scale(self, candidate_id: int) -> torch.Tensor:
        """The candidate_id-th scale of the grid: (alpha, beta) pairs inside each pair of spans."""
        return self._candidate_table()[candidate_id]

    def ask(self) -> torch.Tensor:
        self.candidate = self._candidate_scale(self.candidate_id)
        return self.candidate

    @staticmethod
    def _update_best(*, best_error: list[torch.Tensor] | None, best_scale: torch.Tensor | None,
                     error: list[torch.Tensor], scale: torch.Tensor, numel: int, num_channels: int, num_heads: int,
                     num_head_repeats: int) -> tuple[list[torch.Tensor], torch.Tensor]:
        """smooth.py:655-714.  One error per candidate: it replaces the best when it is not worse on every error
        (ties move on).  One error per group of input channels: a group takes the candidate's scale where the
        candidate is strictly better on every error; with grouped-query heads the errors of the heads that share a
        key/value head are added first, so those heads switch together."""
        if best_error is None:
            return error, scale
        # (selections stay on the device: a host-side comparison would drain the stream once per candidate)
        if numel == 1:
            take = torch.stack([(e <= b).reshape(()) for b, e in zip(best_error, error, strict=True)]).all()
            return ([torch.where(take, e, b) for b, e in zip(best_error, error, strict=True)],
                    torch.where(take, scale, best_scale))
        assert num_channels % numel == 0
        group_size, num_groups = num_channels // numel, numel
        shared = num_heads > 1 and num_head_repeats > 1
        if shared:
            head_channels, unique_heads = num_channels // num_heads, num_heads // num_head_repeats
            groups_per_head, repeats, unique_per_group = 1, 1, 1
            if head_channels >= group_size:
                assert head_channels % group_size == 0
                groups_per_head, repeats = head_channels // group_size, num_head_repeats
            else:
                assert group_size % head_channels == 0
                heads_per_group = group_size // head_channels
                if heads_per_group < num_head_repeats:
                    assert num_head_repeats % heads_per_group == 0
                    repeats = num_head_repeats // heads_per_group
                else:
                    assert heads_per_group % num_head_repeats == 0
                    unique_per_group = heads_per_group // num_head_repeats
            uniques = unique_heads // unique_per_group
            shared = repeats > 1
        pos = torch.full((numel,), True, device=error[0].device)
        for b, e in zip(best_error, error, strict=True):
            if shared:
                b = b.view(uniques, repeats, groups_per_head).sum(dim=1, keepdim=True)
                e = e.view(uniques, repeats, groups_per_head).sum(dim=1, keepdim=True)
                pos = pos & (e < b).expand(uniques, repeats, groups_per_head).reshape_as(pos)
            else:
                pos = pos & (e < b)
        best_error = [torch.where(pos.view_as(b), e, b) for b, e in zip(best_error, error, strict=True)]
        pos = pos.view(num_groups, 1).expand(num_groups, group_size)
        best_scale = torch.where(pos, scale.view(num_groups, group_size), best_scale.view(num_groups, group_size))
        return best_error, best_scale

    def tell(self, error: list[torch.Tensor]) -> None:
        numel = error[0].numel()
        assert all(e.numel() == numel for e in error)
        self.error_history.append([e.clone() for e in error])
        self.best_error, self.best_scale = self._update_best(
            best_error=self.best_error, best_scale=self.best_scale, error=error, scale=self.candidate, numel=numel,
            num_channels=self.num_in_channels, num_heads=self.num_heads, num_head_repeats=self.num_head_repeats)
        self.candidate_id += 1
        if self.candidate_id >= self.population_size:
            self.iter += 1
            self.candidate_id = 0

    # ---- candidate processing
    def _process_x_in_xw(self, x: torch.Tensor, channels_dim=None) -> torch.Tensor:
        """x -> Q(x / s) * s in the develop dtype, back in x's dtype (smooth.py:504-517)."""
        if not self.needs_x_quant_for_wgts:
            return x
        channels_dim = self.x_quantizer.channels_dim if channels_dim is None else channels_dim
        dtype = x.dtype
        if self.develop_dtype == torch.float32 and channels_dim in (-1, x.ndim - 1):
            # the three steps in one pass over x (bit-identical), when the quantizer's configuration allows it
            fused = self.x_quantizer.smooth_candidates(x, self.candidate, divide_first=True, channels_dim=channels_dim,
                                                       default_dtype=dtype)
            if fused is not None:
                return fused
        xs = scale_channels(x, self.candidate, channels_dim, divide=True, out_dtype=self.develop_dtype)
        xq = self.x_quantizer.quantize(xs, channels_dim=channels_dim, default_dtype=dtype,
                                       develop_dtype=self.develop_dtype).data
        return scale_channels(xq, self.candidate, channels_dim, divide=False, out_dtype=dtype).view(x.shape)

    def _process_w_in_xw(self, w: torch.Tensor) -> torch.Tensor:
        """w -> Q(w * s) / s (smooth.py:519-530)."""
        if not self.needs_w_quant_for_wgts:
            return w
        dtype = w.dtype
        channels_dim = 1 if self.w_quantizer.channels_dim is None else self.w_quantizer.channels_dim
        if self.develop_dtype == torch.float32 and w.ndim == 2 and channels_dim == 1:
            fused = self.w_quantizer.smooth_candidates(w, self.candidate, divide_first=False, channels_dim=None,
                                                       default_dtype=dtype)
            if fused is not None:
                return fused
        ws = scale_channels(w, self.candidate, channels_dim, divide=False, out_dtype=self.develop_dtype)
        wq = self.w_quantizer.quantize(ws, kernel=None, default_dtype=dtype, develop_dtype=self.develop_dtype).data
        return scale_channels(wq, self.candidate, channels_dim, divide=True, out_dtype=dtype)

    def _process_wgts_centric_mod(self, wgts: list[nn.Parameter], mods: list[nn.Module],
                                  splits: list[int] | None = None) -> None:
        if self.needs_w_quant_for_wgts and self.config.allow_low_rank and self.w_quantizer.is_enabled_low_rank():
            # smooth.py:588-623: the smoothed weight W s is split into a low-rank branch and a quantized remainder,
            # both fed with x / s (the branch sees it unquantized, the remainder after the input quantizer)
            assert len(wgts) == len(mods)
            for w in wgts:
                self._state_dict.append((w, w.data))
                w.data = scale_channels(w.data, self.candidate, 1, divide=False)
            input_packager = self.x_quantizer.get_input_packager() if self.needs_x_quant else None
            channels_dim = self.x_quantizer.channels_dim if self.x_quantizer is not None else -1
            for mod in mods:
                self._hooks.append(ActivationSmoother(self.candidate, channels_dim, develop_dtype=self.develop_dtype,
                                                      input_packager=input_packager).as_hook().register(mod))
            bounds = [0, *(splits or []), len(wgts)]
            for begin, end in zip(bounds[:-1], bounds[1:]):
                qtensors, branches = self.w_quantizer.quantize_with_low_rank(wgts[begin:end], kernel=None,
                                                                             develop_dtype=self.develop_dtype)
                for qt, branch, w, mod in zip(qtensors, branches, wgts[begin:end], mods[begin:end], strict=True):
                    w.data = qt.data
                    self._hooks.append(branch.as_hook(input_packager=input_packager).register(mod))
                    if self.needs_x_quant_for_wgts:
                        self._hooks.append(self.x_quantizer.as_hook().register(mod))
            return
        if self.needs_w_quant_for_wgts:
            for w in wgts:
                self._state_dict.append((w, w.data))
                w.data = self._process_w_in_xw(w.data)
        if self.needs_x_quant_for_wgts:
            self._hooks.append(self.x_quantizer.as_hook(func=self._process_x_in_xw, is_output=False).register(mods))

    def _reduced_scale(self, scale: torch.Tensor) -> torch.Tensor:
        """The scale of the key/value heads: the first of each group of query heads that share one."""
        if self.num_unique_heads > 0:
            return scale.view(self.num_unique_heads, self.num_head_repeats, -1)[:, 0, :].reshape(-1)
        return scale

    def _quantize_output(self, quantizer: Quantizer, t: torch.Tensor) -> torch.Tensor:
        """smooth.py:532-572 with the OutputsError objective: the producing weights were already scaled, so the
        tensor is only fake-quantized."""
        return quantizer.quantize(t, channels_dim=quantizer.channels_dim, default_dtype=t.dtype,
                                  develop_dtype=self.develop_dtype).data.view(t.shape)

    def _process_opts_centric_mod(self, x_wgts: list[nn.Parameter], y_wgts: list[nn.Parameter],
                                  x_mods: list[nn.Module], y_mods: list[nn.Module]) -> None:
        """Candidate in place for a y-x product (smooth.py:627-653): the rows of the x producers are multiplied by
        the scale, the rows of the y producers divided, and both products fake-quantized as they leave their
        modules."""
        for w in x_wgts:
            self._state_dict.append((w, w.data))
            w.data = scale_channels(w.detach().data, self.candidate, 0, divide=False)
        for w in y_wgts:
            self._state_dict.append((w, w.data))
            w.data = scale_channels(w.detach().data, self._reduced_scale(self.candidate), 0, divide=True)
        if self.needs_x_quant_for_opts:
            self._hooks.append(self.x_quantizer.as_hook(lambda t: self._quantize_output(self.x_quantizer, t),
                                                        is_output=True).register(x_mods))
        if self.needs_y_quant_for_opts:
            self._hooks.append(self.y_quantizer.as_hook(lambda t: self._quantize_output(self.y_quantizer, t),
                                                        is_output=True).register(y_mods))

    def _recover_mod(self) -> None:
        for p, w in self._state_dict:
            p.data = w
        self._state_dict.clear()
        for hook in self._hooks:
            hook.remove()
        self._hooks.clear()

    # ---- search
    def _parse_inputs(self, ipts: TensorsCache | None) -> TensorsCache | None:
        if ipts is None:
            return None
        return TensorsCache({key: ipt.repartition(max_batch_size=self.config.sample_batch_size,
                                                  max_size=self.config.sample_size, standardize=False, reshape=True)
                             for key, ipt in ipts.items()})

    @torch.inference_mode()
    def calibrate(self, x_wgts: list[nn.Parameter], x_acts: TensorsCache | None = None,
                  x_mods: list[nn.Module] | None = None, eval_inputs: TensorsCache | None = None,
                  eval_module: nn.Module | None = None, eval_kwargs: dict[str, tp.Any] | None = None,
                  orig_x_wgts: list[tuple[nn.Parameter, torch.Tensor]] | None = None,
                  orig_eval_inputs: TensorsCache | None = None, y_wgts: list[nn.Parameter] | None = None,
                  y_acts: TensorsCache | None = None, y_mods: list[nn.Module] | None = None,
                  orig_y_wgts: list[tuple[nn.Parameter, torch.Tensor]] | None = None, **kwargs) -> torch.Tensor:
        assert all(isinstance(w, nn.Parameter) for w in x_wgts), "wgts should be a list of nn.Parameter"
        objective, granularity = self.config.objective, self.config.granularity
        outputs_centric = self.tensor_type == TensorType.Outputs
        if outputs_centric:  # search.py:363-366
            objective, granularity = SearchBasedCalibObjective.OutputsError, SearchBasedCalibGranularity.Layer
        if objective == SearchBasedCalibObjective.OutputsError:
            assert eval_inputs is not None, "eval_inputs should not be None when objective is OutputsError"
            assert eval_module is not None, "eval_module should not be None when OutputsError"
            if (not outputs_centric and isinstance(eval_module, (nn.Linear, nn.Conv2d))
                    and granularity != SearchBasedCalibGranularity.Layer):
                objective = SearchBasedCalibObjective.ProductsError  # search.py:397-409
            else:
                granularity = SearchBasedCalibGranularity.Layer
        self.objective, self.granularity = objective, granularity
        if objective == SearchBasedCalibObjective.ProductsError:
            return self._calibrate_products(x_wgts, x_acts or eval_inputs, orig_x_wgts,
                                            kwargs.get("orig_x_acts") or orig_eval_inputs)
        if objective == SearchBasedCalibObjective.TensorError:
            return self._calibrate_tensor(x_wgts, x_acts or eval_inputs, orig_x_wgts)
        assert eval_inputs is not None, "eval_inputs should not be None when objective is OutputsError"
        assert eval_module is not None, "eval_module should not be None when OutputsError"
        x_acts = x_acts or eval_inputs
        eval_kwargs = eval_kwargs or {}
        eval_inputs, orig_eval_inputs = self._parse_inputs(eval_inputs), self._parse_inputs(orig_eval_inputs)
        self.reset(x_wgts=x_wgts, x_acts=x_acts, y_wgts=y_wgts, y_acts=y_acts)

        def forward(cache: TensorsCache, index: int) -> torch.Tensor:
            ipt = cache.extract(index, eval_kwargs)
            y = eval_module(*ipt.args, **ipt.kwargs)
            y = y[0] if not isinstance(y, torch.Tensor) else y
            assert isinstance(y, torch.Tensor), "eval_mod should return a tensor"
            return y

        # the outputs to match: original weights (if given) on the original inputs (search.py:746-765, 1011-1042)
        saved = []
        for originals in (orig_x_wgts, orig_y_wgts):
            if originals is not None:
                saved += [(p, p.data) for p, _ in originals]
                for p, w in originals:
                    p.data = w.to(device=p.data.device)
        ref_inputs = orig_eval_inputs or eval_inputs
        num_batches = len(ref_inputs.front().data)
        orig_opts = [forward(ref_inputs, i) for i in range(num_batches)]
        for p, w in saved:
            p.data = w
        # the evaluated module is the smoothed linear itself: every candidate from codes on the int8 matrix cores
        table = None if outputs_centric else self._all_candidates_linear_outputs_error(
            x_wgts, x_mods, eval_module, eval_inputs, eval_kwargs, orig_opts)
        while not self.is_done():
            self.ask()
            if table is not None:
                self.tell([table[self.candidate_id].view(1)])
                continue
            if outputs_centric:
                self._process_opts_centric_mod(x_wgts=x_wgts, y_wgts=y_wgts, x_mods=x_mods, y_mods=y_mods)
            else:
                self._process_wgts_centric_mod(wgts=x_wgts, mods=x_mods, splits=kwargs.get("splits"))
            total = None
            for i in range(len(eval_inputs.front().data)):
                y = (forward(eval_inputs, i) - orig_opts[i].to(x_wgts[0].device)).to(self.develop_dtype)
                y = y.pow_(self.config.degree).sum().view(-1)
                total = y if total is None else total.add_(y)
            self._recover_mod()
            if self.reduce_sum is not None:
                total = self.reduce_sum(total)
            self.tell([total])
        return self.get_best()


    # bytes of int8 codes one call of the fused output scorer may hold (the candidates are taken in runs that fit)
    LINEAR_SCORER_CODE_BYTES = 2 << 30

    def _all_candidates_linear_outputs_error(self, x_wgts: list[nn.Parameter], x_mods: list[nn.Module] | None,
                                             eval_module: nn.Module, eval_inputs: TensorsCache, eval_kwargs: dict,
                                             orig_opts: list[torch.Tensor]) -> torch.Tensor | None:
        """OutputsError of every candidate, float32 [candidates], when the evaluated module is the one smoothed
        linear (smooth.py:575-625 puts Q(w s) / s into the module and Q(x / s) s in front of it, search.py:1011-1042
        sums (y - y0)^2): s cancels in x' w'^T, so a candidate is one pass that writes int8 codes of w and of each
        batch and an integer GEMM whose epilogue takes the difference from the outputs to match -- neither the
        processed tensors nor the outputs are written.  None when the case is outside that path (another module
        around the linear, a low-rank branch, hooks on the module, 16-bit tensors whose processed values the
        reference rounds to 16 bits, degree other than 2, quantizers outside the fused candidate kernel): the
        caller then evaluates the module candidate by candidate."""
        if (self.num_iters != 1 or self.develop_dtype != torch.float32 or float(self.config.degree) != 2.0
                or not (self.needs_w_quant_for_wgts and self.needs_x_quant_for_wgts)):
            return None
        if self.config.allow_low_rank and self.w_quantizer.is_enabled_low_rank():
            return None
        if (not isinstance(eval_module, nn.Linear) or x_mods is None or len(x_mods) != 1 or x_mods[0] is not eval_module
                or len(x_wgts) != 1 or x_wgts[0] is not eval_module.weight or eval_kwargs
                or eval_module._forward_pre_hooks or eval_module._forward_hooks):
            return None
        w = x_wgts[0].data
        if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous() or eval_inputs.num_tensors != 1:
            return None
        if self.x_quantizer.get_input_packager() is not None or self.x_quantizer.channels_dim not in (-1, None):
            return None
        bias = eval_module.bias
        if bias is not None and bias.dtype != torch.float32:
            return None
        batches = eval_inputs.front().data
        R, Cn = w.shape
        if (Cn % 16 or len(batches) != len(orig_opts)
                or any(x.dtype != torch.float32 or x.shape[-1] != Cn for x in batches)
                or any(y.dtype != torch.float32 or y.shape[-1] != R for y in orig_opts)):
            return None
        scales = self._candidate_table().to(torch.float32)
        K = scales.shape[0]
        rows = max(x.numel() // Cn for x in batches)
        run = max(1, min(K, self.LINEAR_SCORER_CODE_BYTES // ((R + rows) * Cn)))
        totals = []
        for begin in range(0, K, run):
            cs = scales[begin:begin + run]
            w_codes = self.w_quantizer.smooth_candidates(w, cs, divide_first=False, channels_dim=None,
                                                         default_dtype=w.dtype, as_codes=True)
            if w_codes is None:
                return None
            total = None
            for x, y0 in zip(batches, orig_opts, strict=True):
                x = x.to(w.device).contiguous()
                x_codes = self.x_quantizer.smooth_candidates(x, cs, divide_first=True, channels_dim=-1,
                                                             default_dtype=x.dtype, as_codes=True)
                if x_codes is None:
                    return None
                gx, gw = Cn // x_codes[1].shape[-1], Cn // w_codes[1].shape[-1]
                if (min(gx, gw) < Cn and (max(gx, gw) % min(gx, gw) or min(gx, gw) % 128)) or min(gx, gw) > (1 << 17):
                    return None
                total = engine.linear_candidate_errors(x_codes[0], x_codes[1], w_codes[0], w_codes[1], bias,
                                                       y0.to(w.device).reshape(-1, R), out=total)
                del x_codes
            totals.append(total)
        errors = torch.cat(totals).to(torch.float32)
        return errors if self.reduce_sum is None else self.reduce_sum(errors)

    def _calibrate_tensor(self, x_wgts: list[nn.Parameter], x_acts: TensorsCache | None,
                          orig_x_wgts: list[tuple[nn.Parameter, torch.Tensor]] | None) -> torch.Tensor:
        """TensorError (search.py:690-696, 776-790): every candidate is scored on the processed weights against the
        originals, |Q(w s) / s - w|^degree evaluated in the weights' dtype and summed per group of input channels over
        all rows (ChannelGroup) or over the whole weight (Layer).  The kernel hands back the unrounded float32 sum of
        every (row, group); they are added up in float64 and rounded to the dtype once, as the reference's one
        multi-axis ``sum`` rounds once."""
        assert self.w_quantizer is not None, "TensorError takes the group geometry from the weight quantizer"
        wgts = [w.detach().data for w in x_wgts]
        origs = wgts if orig_x_wgts is None else [w for _, w in orig_x_wgts]
        assert len(origs) == len(wgts)
        assert all(w.shape[1:] == wgts[0].shape[1:] for w in wgts + origs)
        views = [infer_view_shape(w.shape, self.w_quantizer.config.largest_group_shape) for w in wgts]
        refs = [engine.canonical_groups(o.to(device=w.device, dtype=w.dtype), view)[0]
                for o, w, view in zip(origs, wgts, views)]
        self.reset(x_wgts=x_wgts, x_acts=x_acts)
        degree = float(self.config.degree)
        # every candidate of every weight from ONE read of the weight, when the fused kernel covers the configuration
        # (then the candidates' errors are looked up below instead of being computed one candidate at a time)
        table = self._all_candidates_tensor_error(wgts, origs, degree)
        while not self.is_done():
            self.ask()
            errors = []
            for index, (w, view, ref) in enumerate(zip(wgts, views, refs, strict=True)):
                if table is not None:
                    raw = table[index][self.candidate_id + self.iter * self.population_size].view(view[::2]).double()
                else:
                    a2d, _ = engine.canonical_groups(self._process_w_in_xw(w), view)
                    raw = engine.group_error(a2d, ref, degree).view(view[::2]).double()
                if self.granularity == SearchBasedCalibGranularity.Layer:
                    total = raw.sum().view(-1)
                else:  # one error per group of input channels (Group is turned into ChannelGroup by the config)
                    total = raw.sum(dim=[d for d in range(raw.ndim) if d != 1]).view(view[2])
                errors.append(total.to(w.dtype))  # (weights are whole on every process: nothing to reduce)
            self.tell(errors)
        return self.get_best()

    def _all_candidates_tensor_error(self, wgts: list[torch.Tensor], origs: list[torch.Tensor], degree: float):
        """Per weight the float32 [candidates][groups] table of |Q(w s) / s - w|^degree sums, or None when a weight is
        compared with something other than itself, or the quantizer is outside the fused kernel."""
        if (not self.needs_w_quant_for_wgts or self.develop_dtype != torch.float32 or self.num_iters != 1
                or any(o is not w for o, w in zip(origs, wgts))):
            return None
        channels_dim = 1 if self.w_quantizer.channels_dim is None else self.w_quantizer.channels_dim
        if channels_dim != 1 or any(w.ndim != 2 for w in wgts):
            return None
        # (the candidate list does not depend on the errors)
        scales = self._candidate_table().to(torch.float32)
        tables = []
        for w in wgts:
            errors = self.w_quantizer.smooth_candidates(w, scales, divide_first=False, channels_dim=None,
                                                        default_dtype=w.dtype, errors_degree=degree)
            if errors is None:
                return None
            tables.append(errors)
        return tables

    def _calibrate_products(self, x_wgts: list[nn.Parameter], x_acts: TensorsCache | None,
                            orig_x_wgts: list[tuple[nn.Parameter, torch.Tensor]] | None,
                            orig_x_acts: TensorsCache | None) -> torch.Tensor:
        """ProductsError (search.py:697-745, 791-816): every candidate is scored on the products of the processed
        input batches with the processed weights against the products of the originals, in the weights' dtype.
        Layer granularity takes the whole product x @ w^T; ChannelGroup / Group take the partial products of each
        quantization group of input channels (one batched GEMM over the groups here), summed per channel group or
        kept per (row, group)."""
        assert x_acts is not None, "x_acts should not be None when objective is ProductsError"
        assert self.w_quantizer is not None, "ProductsError takes the group geometry from the weight quantizer"
        wgts = [w.detach().data for w in x_wgts]
        origs = wgts if orig_x_wgts is None else [w for _, w in orig_x_wgts]
        assert len(origs) == len(wgts)
        if any(w.ndim != 2 for w in wgts):
            raise NotImplementedError("ProductsError smoothing search is implemented for 2-D (linear) weights")
        views = [infer_view_shape(w.shape, self.w_quantizer.config.largest_group_shape) for w in wgts]
        if any(v[1] != 1 for v in views):
            raise NotImplementedError("ProductsError smoothing search needs (1, g) weight groups")
        layer = self.granularity == SearchBasedCalibGranularity.Layer
        dev, dt = wgts[0].device, wgts[0].dtype

        def parse(cache: TensorsCache | None) -> TensorsCache | None:
            if cache is None:
                return None
            return TensorsCache({key: ipt.repartition(max_batch_size=self.config.element_batch_size,
                                                      max_size=self.config.element_size, standardize=True, reshape=True)
                                 for key, ipt in cache.items()})

        ipts, orig_ipts = parse(x_acts), parse(orig_x_acts)
        self.reset(x_wgts=x_wgts, x_acts=x_acts)
        assert ipts.num_tensors == 1, "linears that share a smoothing scale take one input"
        cache = ipts.front()

        def products(x: torch.Tensor, w: torch.Tensor, view) -> torch.Tensor:
            """x [n, C] and w [R, C] -> [n, R] (Layer) or [CG * R, n] (partial products per channel group)."""
            x = cache.reshape(x).view(x.shape[0], -1).to(device=dev, dtype=dt)
            # float64 accumulation, rounded once (engine.grouped_products): no float32 summation order to depend on
            if layer:
                wk = w.view(w.shape[0], -1).t().contiguous()
                return engine.grouped_products(x, wk.view(1, *wk.shape), dt).view(x.shape[0], -1)
            R, _, CG, g = view
            xg = x.view(-1, CG, g).permute(1, 2, 0).contiguous()
            return engine.grouped_products(w.view(R, CG * g), xg, dt).view(CG * R, -1)

        orig_cache = cache if orig_ipts is None else orig_ipts.front()
        base = [[products(x, w, view) for x in orig_cache.data] for w, view in zip(origs, views)]
        degree = float(self.config.degree)
        while not self.is_done():
            self.ask()
            batches = [self._process_x_in_xw(x.to(dev), channels_dim=cache.channels_dim) for x in cache.data]
            errors = []
            for w, view, y0s in zip(wgts, views, base, strict=True):
                qw = self._process_w_in_xw(w)
                total = None
                for x, y0 in zip(batches, y0s, strict=True):
                    total = engine.product_error(products(x, qw, view), y0, degree, out=total)
                if layer:
                    total = total.double().sum().to(torch.float32).view(1)
                else:
                    total = total.view(view[2], view[0])
                    if self.granularity == SearchBasedCalibGranularity.ChannelGroup:
                        total = total.double().sum(dim=1).to(torch.float32)
                    else:
                        total = total.t().contiguous()
                errors.append(total if self.reduce_sum is None else self.reduce_sum(total.contiguous()))
            self.tell(errors)
        return self.get_best()


class SmoothLinearCalibrator(SmoothCalibrator):
    """Smoothing search for linear modules fed by the same activation (smooth.py:717-754)."""

    def __init__(self, config: SmoothCalibConfig, weight_quantizer: Quantizer | None, input_quantizer: Quantizer | None,
                 num_heads: int = 1, num_head_repeats: int = 1, develop_dtype: torch.dtype = torch.float32) -> None:
        super().__init__(tensor_type=TensorType.Weights, config=config, w_quantizer=weight_quantizer,
                         x_quantizer=input_quantizer, y_quantizer=None, num_heads=num_heads,
                         num_head_repeats=num_head_repeats, develop_dtype=develop_dtype)


class SmoothAttentionCalibrator(SmoothCalibrator):
    """Query / key smoothing search (smooth.py:757-849): the scale multiplies the query projection's rows and
    divides the key projection's, scored on the outputs of the attention module."""

    def __init__(self, config: SmoothCalibConfig, query_quantizer: Quantizer | None, key_quantizer: Quantizer | None,
                 num_heads: int = 1, num_head_repeats: int = 1, with_rope: bool = True,
                 develop_dtype: torch.dtype = torch.float32) -> None:
        super().__init__(tensor_type=TensorType.Outputs, config=config, w_quantizer=None, x_quantizer=query_quantizer,
                         y_quantizer=key_quantizer, num_heads=num_heads, num_head_repeats=num_head_repeats,
                         with_rope=with_rope, develop_dtype=develop_dtype)

    def calibrate(self, q_proj_weight: nn.Parameter, k_proj_weight: nn.Parameter, queries: TensorsCache,
                  keys: TensorsCache, query_module: nn.Module, key_module: nn.Module,
                  eval_module: nn.Module | None = None, eval_inputs: TensorsCache | None = None,
                  eval_kwargs: dict[str, tp.Any] | None = None) -> torch.Tensor:
        return super().calibrate(x_wgts=[q_proj_weight], y_wgts=[k_proj_weight], x_acts=queries, y_acts=keys,
                                 x_mods=[query_module], y_mods=[key_module], eval_module=eval_module,
                                 eval_inputs=eval_inputs, eval_kwargs=eval_kwargs)


def _check_finite(param: nn.Parameter) -> None:
    assert not param.data.isnan().any(), "NaN found in param when smoothing"
    assert not param.data.isinf().any(), "Inf found in param when smoothing"


def smooth_upscale_param(param: nn.Parameter, scale: torch.Tensor, channels_dim: int = 1) -> None:
    """param <- (param.to(scale.dtype) * scale).to(param.dtype) along ``channels_dim`` (smooth.py:852-869)."""
    if scale.dtype != torch.float32:
        shape = [1] * param.ndim
        shape[channels_dim] = -1
        param.data = param.data.to(dtype=scale.dtype).mul_(scale.to(param.device).view(shape)).to(dtype=param.dtype)
    else:
        param.data = scale_channels(param.data, scale, channels_dim, divide=False)
    _check_finite(param)


def smooth_downscale_param(param: nn.Parameter, scale: torch.Tensor, channels_dim: int = 0) -> None:
    """The first ``scale.numel()`` channels of param are divided by scale (smooth.py:872-891)."""
    if scale.dtype != torch.float32:
        shape = [1] * param.ndim
        shape[channels_dim] = -1
        data = param.data.to(dtype=scale.dtype)
        data.narrow(channels_dim, 0, scale.numel()).div_(scale.to(param.device).view(shape))
        param.data = data.to(dtype=param.dtype)
    else:
        param.data = scale_channels(param.data, scale, channels_dim, divide=True)
    _check_finite(param)


def convert_smooth_upscale_to_downscale(scale: torch.Tensor, num_heads: int = 1, num_head_repeats: int = 1) -> torch.Tensor:
    """With grouped-query attention the producer has one head per group of repeated heads (smooth.py:894-916)."""
    if num_heads > 1 and num_head_repeats > 1:
        head_channels = scale.numel() // num_heads
        return scale.view(num_heads // num_head_repeats, num_head_repeats, head_channels)[:, 0, :].reshape(-1)
    return scale


@torch.inference_mode()
def smooth_linear_modules(prevs: nn.Module | tp.Sequence[nn.Module] | None, modules: tp.Sequence[nn.Linear] | nn.Linear,
                          *, scale: torch.Tensor | None, config: SmoothCalibConfig | None = None,
                          weight_quantizer: Quantizer | None = None, input_quantizer: Quantizer | None = None,
                          weights: list[nn.Parameter] | None = None, inputs: TensorsCache | None = None,
                          eval_inputs: TensorsCache | None = None, eval_module: nn.Module = None,
                          eval_kwargs: dict[str, tp.Any] = None, num_heads: int = 1, num_head_repeats: int = 1,
                          splits: list[int] | None = None, extra_modules: list[nn.Linear] | None = None,
                          develop_dtype: torch.dtype = torch.float32, world: int = 1) -> torch.Tensor:
    """Find (if ``scale`` is None) and apply the smoothing scale between ``prevs`` and ``modules``: the modules'
    input channels are multiplied, the producers' output channels divided.  Returns the scale on the CPU
    (smooth.py:919-1015).  ``world`` > 1: the cached activations are this rank's shard of the calibration tokens; the
    activation spans and the candidates' errors are all-reduced over the default process group."""
    modules = list(modules) if isinstance(modules, (list, tuple)) else [modules]
    extra_modules = [] if extra_modules is None else list(extra_modules)
    if scale is None:
        assert inputs is not None or eval_inputs is not None, "inputs or eval_inputs must be provided"
        calibrator = SmoothLinearCalibrator(config=config, weight_quantizer=weight_quantizer,
                                            input_quantizer=input_quantizer, num_heads=num_heads,
                                            num_head_repeats=num_head_repeats, develop_dtype=develop_dtype)
        if world > 1:
            from ..parallel import all_reduce_max, all_reduce_sum

            calibrator.reduce_max = lambda t: all_reduce_max(t, world)
            calibrator.reduce_sum = lambda t: all_reduce_sum(t, world)
        scale = calibrator.calibrate(
            x_wgts=[m.weight for m in modules] if weights is None else weights, x_acts=inputs, x_mods=modules,
            eval_inputs=eval_inputs, eval_module=eval_module, eval_kwargs=eval_kwargs, splits=splits)
        gc.collect()
        torch.cuda.empty_cache()
    upscale = scale
    for module in modules + extra_modules:
        upscale = upscale.to(device=module.weight.device)
        smooth_upscale_param(module.weight, upscale, channels_dim=1)
    if prevs is not None:
        downscale = convert_smooth_upscale_to_downscale(upscale, num_heads=num_heads, num_head_repeats=num_head_repeats)
        for module in [prevs] if isinstance(prevs, nn.Module) else prevs:
            if module is None:
                continue
            downscale = downscale.to(device=module.weight.device)
            smooth_downscale_param(module.weight, downscale, channels_dim=0)
            if getattr(module, "bias", None) is not None:
                smooth_downscale_param(module.bias, downscale, channels_dim=0)
    return scale.to(device="cpu")


@torch.inference_mode()
def smooth_attention(*, q_proj: nn.Linear, k_proj: nn.Linear, scale: torch.Tensor | None,
                     config: SmoothCalibConfig | None = None, query_quantizer: Quantizer | None = None,
                     key_quantizer: Quantizer | None = None, queries: TensorsCache | None = None,
                     keys: TensorsCache | None = None, attn_q: nn.Module | None = None,
                     attn_k: nn.Module | None = None, eval_inputs: TensorsCache | None = None,
                     eval_module: nn.Module = None, eval_kwargs: dict[str, tp.Any] = None, num_heads: int = 1,
                     num_head_repeats: int = 1, with_rope: bool = True,
                     develop_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Find (if ``scale`` is None) and apply the query / key smoothing scale: q_proj rows are multiplied, k_proj
    rows divided.  Returns the scale on the CPU (smooth.py:1018-1109)."""
    if scale is None:
        assert queries is not None and keys is not None and eval_inputs is not None
        assert attn_q is not None and attn_k is not None, "modules must be provided"
        scale = SmoothAttentionCalibrator(
            config=config, query_quantizer=query_quantizer, key_quantizer=key_quantizer, num_heads=num_heads,
            num_head_repeats=num_head_repeats, with_rope=with_rope, develop_dtype=develop_dtype).calibrate(
            q_proj_weight=q_proj.weight, k_proj_weight=k_proj.weight, queries=queries, keys=keys,
            query_module=attn_q, key_module=attn_k, eval_inputs=eval_inputs, eval_module=eval_module,
            eval_kwargs=eval_kwargs)
        gc.collect()
        torch.cuda.empty_cache()
    upscale = scale.to(device=q_proj.weight.device)
    smooth_upscale_param(q_proj.weight, upscale, channels_dim=0)
    downscale = convert_smooth_upscale_to_downscale(upscale, num_heads=num_heads, num_head_repeats=num_head_repeats)
    smooth_downscale_param(k_proj.weight, downscale, channels_dim=0)
    return scale.to(device="cpu")
