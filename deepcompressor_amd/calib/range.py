"""Dynamic-range calibration (reference: deepcompressor/calib/range.py:25-474, calib/search.py:63-837).

The reference proposes one ratio at a time (ask), fake-quantizes the whole tensor, measures the error and
keeps the best (tell).  Two device paths produce the same result here:

* fused -- single-step quantizer, one scale level, groups of at most 512 contiguous values: ONE kernel launch
  evaluates every candidate of every group from registers (``dcq_range_search_groups``); Group granularity
  takes the per-group winners straight from the kernel, ChannelGroup / Layer add the per-candidate group sums
  up over rows / the whole tensor first;
* general -- anything else the weight search covers (a step of a progressive quantizer with the previous
  steps' scale, multi-level scales, long or non-contiguous groups, and the ProductsError / OutputsError
  objectives): the ask/tell loop itself, each candidate fake-quantized by the HIP quantizer and scored by
  ``dcq_group_error`` (the weight itself), ``dcq_product_error`` over the per-group partial products x @ w
  (AWQ-style clipping search) or over the evaluation module's outputs.
"""

from __future__ import annotations

import typing as tp

import torch
import torch.nn as nn

from .. import engine
from ..data.cache import TensorsCache
from ..data.common import TensorType
from ..data.range import DynamicRange
from ..data.scale import QuantScale
from ..data.utils.shape import infer_view_shape
from ..quantizer.impl.info import QuantInfo
from ..quantizer.processor import Quantizer
from .config import (DynamicRangeCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective)

__all__ = ["DynamicRangeCalibrator", "calibrate_dynamic_range"]


class DynamicRangeCalibrator:
    """Finds the (min, max) per group / channel group / tensor that minimises the quantization error of a tensor."""

    def __init__(self, tensor_type: TensorType, config: DynamicRangeCalibConfig, static: bool, quantizer: Quantizer,
                 pre_scale: torch.Tensor | None = None, reduce_errors=None) -> None:
        assert quantizer is not None and quantizer.is_enabled(), "The tensor should be quantized."
        self.tensor_type, self.config, self.static, self.quantizer = tensor_type, config, static, quantizer
        self.pre_scale = pre_scale
        # rows of the weight sharded over processes: errors that are summed over rows (ChannelGroup / Layer
        # granularity) are partial here; ``reduce_errors(float64 tensor) -> tensor`` adds them up over the shards
        # (an all-reduce) before they are rounded and compared, so every shard takes the same decision
        self.reduce_errors = reduce_errors
        self.develop_dtype = quantizer.develop_dtype
        self.objective, self.granularity = config.objective, config.granularity
        self.ratios = config.get_ratios()
        self.num_iters = len(self.ratios)
        self.base_range = DynamicRange()
        self.best_range = DynamicRange()
        self.best_error: torch.Tensor | None = None
        self.candidate_errors: torch.Tensor | None = None
        self.path: str | None = None  # "fused" or "general", for tests and logs

    def is_clamp_based(self) -> bool:
        return self.static or not self.config.allow_scale

    def _unsupported(self, what: str):
        return NotImplementedError(
            f"range calibration with {what} is not implemented on the device "
            "(implemented: weights, TensorError objective, static / clamp-based range, any granularity)")

    # ---- shapes (calib/range.py:113-139) ---------------------------------------------------------------------
    def _set_shapes(self, shape: torch.Size) -> None:
        self._set_shapes_from_view(infer_view_shape(shape, self.quantizer.config.largest_group_shape), shape)

    def _set_shapes_from_view(self, view: torch.Size, shape: torch.Size | None = None) -> None:
        self.view_shape = view
        self.pos_view_shape = torch.Size([1, 1, view[2], *([1] * (len(view) - 3))])
        self.range_shape = torch.Size([gs if i % 2 == 0 else 1 for i, gs in enumerate(view)])
        if self.granularity == SearchBasedCalibGranularity.Layer:
            self.ratio_shape = self.error_shape = torch.Size((1,))
            self.ratio_view_shape = self.ratio_shape
        elif self.granularity == SearchBasedCalibGranularity.ChannelGroup:
            self.ratio_shape = self.error_shape = torch.Size((view[2],))
            self.ratio_view_shape = self.pos_view_shape
        elif self.granularity == SearchBasedCalibGranularity.Group:
            self.ratio_shape = self.error_shape = torch.Size(view[::2])
            self.ratio_view_shape = self.range_shape
        else:
            raise ValueError(f"Invalid granularity: {self.granularity}")
        if self.pre_scale is not None:
            assert len(shape) * 2 == len(self.pre_scale.shape)
            self.pre_view_shape = infer_view_shape(shape, self.pre_scale.shape[1::2])
        else:
            self.pre_view_shape = torch.Size()

    def _reduce(self, raw: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """Unrounded float32 group sums shaped (..., *view[::2]) -> the error tensor of this granularity in the
        weight's dtype (calib/search.py:780-787: the reference's sum runs in that dtype, accumulating in float32
        and rounding once; groups are added up in float64 here, so a Layer / ChannelGroup error can differ from
        the reference's by one unit in the last place of ``dtype`` when the float32 total sits on a rounding edge)."""
        lead = raw.ndim - len(self.view_shape[::2])
        if self.granularity == SearchBasedCalibGranularity.Group:
            return raw.to(dtype)
        dims = [lead + i for i in range(len(self.view_shape[::2]))
                if not (self.granularity == SearchBasedCalibGranularity.ChannelGroup and i == 1)]
        total = raw.double().sum(dim=dims)
        if self.reduce_errors is not None:
            total = self.reduce_errors(total)
        return total.view(*raw.shape[:lead], *self.error_shape).to(dtype)

    # ---- entry -------------------------------------------------------------------------------------------------
    def _resolve_objective(self, eval_module) -> None:
        """calib/search.py:361-414: OutputsError on a single linear / conv below Layer granularity is scored on
        partial products; any other OutputsError is one scalar per candidate."""
        self.objective, self.granularity = self.config.objective, self.config.granularity
        if self.objective == SearchBasedCalibObjective.OutputsError:
            if (isinstance(eval_module, (nn.Linear, nn.Conv2d))
                    and self.granularity.value < SearchBasedCalibGranularity.Layer.value):
                self.objective = SearchBasedCalibObjective.ProductsError
            else:
                self.granularity = SearchBasedCalibGranularity.Layer

    @torch.inference_mode()
    def calibrate(self, x_wgts: list[nn.Parameter] | None = None, *, y_wgts: list[nn.Parameter] | None = None,
                  x_acts: TensorsCache | None = None, y_acts: TensorsCache | None = None,
                  x_mods: list[nn.Module] | None = None, y_mods: list[nn.Module] | None = None,
                  eval_inputs: TensorsCache | None = None,
                  eval_module: nn.Module | None = None, eval_kwargs: dict[str, tp.Any] | None = None,
                  orig_x_wgts: list[tuple[nn.Parameter, torch.Tensor]] | None = None,
                  orig_y_wgts: list[tuple[nn.Parameter, torch.Tensor]] | None = None,
                  orig_x_acts: TensorsCache | None = None, orig_y_acts: TensorsCache | None = None,
                  orig_eval_inputs: TensorsCache | None = None,
                  record_errors: bool = False, **unused) -> DynamicRange:
        """Run the whole grid search (SearchBasedCalibrator.calibrate, calib/search.py:532-684)."""
        if self.tensor_type != TensorType.Weights:
            inputs = self.tensor_type == TensorType.Inputs
            self.path = "general"
            self._calibrate_acts(acts=x_acts if inputs else y_acts, mods=x_mods if inputs else y_mods,
                                 wgts=x_wgts if inputs else None, orig_acts=orig_x_acts if inputs else orig_y_acts,
                                 eval_inputs=eval_inputs, eval_module=eval_module, eval_kwargs=eval_kwargs or {},
                                 orig_wgts=(orig_x_wgts or []) + (orig_y_wgts or []),
                                 orig_eval_inputs=orig_eval_inputs, record_errors=record_errors)
            return self.get_best()
        if not self.is_clamp_based():
            raise self._unsupported("allow_scale on a non-static tensor")
        assert x_wgts is not None and len(x_wgts) == 1, "The weight should be a single tensor."
        self._resolve_objective(eval_module)
        param = x_wgts[0]
        w = param.detach().data
        self._set_shapes(w.shape)
        q = self.quantizer
        info = q.update(w.shape, default_dtype=w.dtype)  # the search fake-quantizes with default_dtype = w.dtype
        if self.objective == SearchBasedCalibObjective.ProductsError:
            x_acts = x_acts or eval_inputs
            assert x_acts is not None, "x_acts should not be None when objective is ProductsError"
            self.path = "general"
            self._calibrate_general(w, record_errors, self._products_scorer(w, x_acts, orig_x_wgts,
                                                                            orig_x_acts or orig_eval_inputs))
            return self.get_best()
        if self.objective == SearchBasedCalibObjective.OutputsError:
            assert eval_inputs is not None, "eval_inputs should not be None when objective is OutputsError"
            assert eval_module is not None, "eval_module should not be None when OutputsError"
            assert isinstance(param, nn.Parameter), "wgts should be a list of nn.Parameter"
            self.path = "general"
            self._calibrate_general(w, record_errors, self._outputs_scorer(
                param, eval_inputs, eval_module, eval_kwargs or {}, orig_x_wgts, orig_eval_inputs))
            return self.get_best()
        group_size = 1
        for s in self.view_shape[1::2]:
            group_size *= s
        fp_with_zero = q.config.quant_dtype.is_float_point and q.config.zero_domain is not None
        fused = (self.pre_scale is None and info.num_steps == 1 and info.steps[0].scale.is_single_level()
                 and not fp_with_zero
                 and tuple(info.steps[0].tensor_view_shape) == tuple(self.view_shape)
                 and group_size in (16, 32, 64, 128, 256, 512)  # the team sizes the fused kernel is built for
                 and not (q.range_bound is not None and q.range_bound.is_set()))
        self.path = "fused" if fused else "general"
        if fused:
            self._calibrate_fused(w, info.steps[0], record_errors)
        else:
            self._calibrate_general(w, record_errors, self._tensor_scorer(w))
        return self.get_best()

    # ---- fused path --------------------------------------------------------------------------------------------
    def _calibrate_fused(self, w: torch.Tensor, step_info, record_errors: bool) -> None:
        view = self.view_shape
        x2d, _ = engine.canonical_groups(w, view)
        ratios = [r for it in self.ratios for r in it]
        group = self.granularity == SearchBasedCalibGranularity.Group
        out = engine.range_search_groups(step_info.scale.native(w.device), x2d, ratios, float(self.config.degree),
                                         want_errors=record_errors or not group, unrounded_errors=not group)
        if group:
            self.best_range = DynamicRange(min=out["min"].view(self.range_shape) if out["min"] is not None else None,
                                           max=out["max"].view(self.range_shape), ratio=out["ratio"].view(view[::2]))
            self.best_error = out["error"].view(view[::2]).to(w.dtype)
            if record_errors:
                self.candidate_errors = out["errors"].view(len(ratios), *view[::2])
            return
        errors = self._reduce(out["errors"].view(len(ratios), *view[::2]), w.dtype)  # [K, *error_shape]
        self.best_error, first = _first_strict_min(errors)
        r32 = engine._device_ratios(tuple(ratios), w.device)
        self.base_range = DynamicRange.construct(
            w.view(view), zero_domain=q_zero_domain(self.quantizer),
            is_float_point=self.quantizer.config.quant_dtype.is_float_point)
        best = self.base_range.scale(r32[first].view(self.ratio_view_shape), q_zero_domain(self.quantizer),
                                     self.quantizer.config.quant_dtype.is_float_point)
        self.best_range = DynamicRange(min=None if best.min is None else best.min.expand(self.range_shape).contiguous(),
                                       max=best.max.expand(self.range_shape).contiguous(),
                                       ratio=r32[first].view(self.ratio_shape))
        if record_errors:
            self.candidate_errors = errors

    # ---- general path: the reference's ask / tell loop --------------------------------------------------------
    def _preprocess_with_pre_scale(self, t: torch.Tensor) -> torch.Tensor:
        """clamp(t / pre_scale, range_bound) in float32, shaped like t (calib/range.py:252-258)."""
        # one group per value of the previous scale: the block of the tensor that value is broadcast over
        block = [dim // n for dim, n in zip(t.shape, self.pre_scale.shape[0::2])]
        x2d, restore = engine.canonical_groups(t, infer_view_shape(t.shape, block))
        rb = self.quantizer.range_bound
        lo = rb.min if rb is not None and rb.min is not None else float("-inf")
        hi = rb.max if rb is not None and rb.max is not None else float("inf")
        outer = self.pre_scale.to(torch.float32).contiguous().view(-1)
        return restore(engine.prescale_groups(x2d, outer, float(lo), float(hi)))

    def _process_wxy(self, tensor: torch.Tensor, candidate: DynamicRange) -> torch.Tensor:
        """The fake-quantized tensor of one candidate, in the tensor's dtype (calib/range.py:260-275)."""
        shape, dtype = tensor.shape, tensor.dtype
        if self.pre_scale is not None:
            tensor = self._preprocess_with_pre_scale(tensor).view(shape)
        tensor = self.quantizer.quantize(tensor, kernel=None, channels_dim=None, dynamic_range=candidate,
                                         default_dtype=dtype, develop_dtype=self.develop_dtype).data
        if self.pre_scale is not None:
            tensor = tensor.view(self.pre_view_shape).mul_(self.pre_scale).to(dtype)
        return tensor.view(shape)

    def _calibrate_general(self, w: torch.Tensor, record_errors: bool, scorer=None) -> None:
        view = self.view_shape
        zd = q_zero_domain(self.quantizer)
        is_fp = self.quantizer.config.quant_dtype.is_float_point
        scorer = scorer or self._tensor_scorer(w)
        t = self._preprocess_with_pre_scale(w) if self.pre_scale is not None else w
        self.base_range = DynamicRange.construct(t.view(view).to(self.develop_dtype), zero_domain=zd, is_float_point=is_fp)
        del t
        self.best_range, self.best_error = DynamicRange(), None
        history = []
        for iteration in self.ratios:
            for ratio in iteration:
                candidate = self.base_range.scale(ratio=ratio, zero_domain=zd, is_float_point=is_fp)
                error = scorer(self._process_wxy(w, candidate))
                if record_errors:
                    history.append(error.clone())
                self._tell_general(error, candidate, ratio, w.device)
        if record_errors:
            self.candidate_errors = torch.stack(history)

    # ---- what a candidate is scored on ------------------------------------------------------------------------
    def _tensor_scorer(self, w: torch.Tensor):
        """TensorError (calib/search.py:776-790): |Q(w) - w|^degree summed per group in the weight's dtype."""
        view = self.view_shape
        b2d, _ = engine.canonical_groups(w, view)

        def score(qw: torch.Tensor) -> torch.Tensor:
            a2d, _ = engine.canonical_groups(qw, view)
            return self._reduce(engine.group_error(a2d, b2d, float(self.config.degree)).view(view[::2]), w.dtype)

        return score

    def _parse_ipts(self, ipts: TensorsCache | None) -> TensorsCache | None:
        """calib/search.py:254-285."""
        if ipts is None:
            return None
        products = self.objective == SearchBasedCalibObjective.ProductsError
        batch = self.config.element_batch_size if products else self.config.sample_batch_size
        size = self.config.element_size if products else self.config.sample_size
        return TensorsCache({key: ipt.repartition(max_batch_size=batch, max_size=size, standardize=products,
                                                  reshape=True) for key, ipt in ipts.items()})

    def _products_scorer(self, w: torch.Tensor, x_acts: TensorsCache, orig_x_wgts, orig_x_acts):
        """ProductsError (calib/search.py:697-745, 791-816).  Per input batch x[n, C] the reference multiplies every
        quantization group on its own, y[r, cg, n] = sum_i x[n, cg*g + i] * w[r, cg*g + i], in the weight's dtype
        (a broadcast matmul over (R, C/g) tiny products); here that is one batched GEMM over the C/g column groups,
        accumulated in float64 and rounded once (``engine.grouped_products``): a product does not depend on any
        backend's float32 summation order.
        Error per (r, cg) = sum_n float32(dtype(y - y0))^degree, batches added up in float32."""
        view = self.view_shape
        if len(view) != 4 or view[1] != 1:
            raise self._unsupported("ProductsError for weights that are not 2-D with (1, g) groups")
        R, _, CG, g = view
        assert x_acts.num_tensors == 1, "a linear takes one input"
        ipts, orig_ipts = self._parse_ipts(x_acts), self._parse_ipts(orig_x_acts)
        dev, dt = w.device, w.dtype

        def grouped(x: torch.Tensor) -> torch.Tensor:  # [n, C] -> [CG, g, n]
            return x.to(device=dev, dtype=dt).view(-1, CG, g).permute(1, 2, 0).contiguous()

        def products(xg: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:  # -> [CG * R, n]
            return engine.grouped_products(weight.view(R, CG * g), xg, dt).view(CG * R, -1)

        batches = [grouped(ipts.front().reshape(x)) for x in ipts.front().data]
        orig_w = orig_x_wgts[0][1].to(dev) if orig_x_wgts is not None else w
        orig_batches = batches if orig_ipts is None else [grouped(orig_ipts.front().reshape(x))
                                                          for x in orig_ipts.front().data]
        base = [products(xg, orig_w) for xg in orig_batches]
        degree = float(self.config.degree)

        def score(qw: torch.Tensor) -> torch.Tensor:
            total = None
            for xg, y0 in zip(batches, base):
                total = engine.product_error(products(xg, qw), y0, degree, out=total)
            per_group = total.view(CG, R).t()
            if self.granularity == SearchBasedCalibGranularity.Group:
                return per_group.contiguous()
            total = per_group.double().sum(dim=0)  # ChannelGroup
            if self.reduce_errors is not None:
                total = self.reduce_errors(total)
            return total.to(torch.float32)

        return score

    def _outputs_scorer(self, param: nn.Parameter, eval_inputs: TensorsCache, eval_module: nn.Module,
                        eval_kwargs: dict, orig_x_wgts, orig_eval_inputs):
        """OutputsError (calib/search.py:746-765, 817-832): the evaluation module's outputs with the candidate
        weight in place against its outputs with the original weight; one scalar per candidate."""
        ipts, orig_ipts = self._parse_ipts(eval_inputs), self._parse_ipts(orig_eval_inputs)
        degree = float(self.config.degree)

        def forward(cache: TensorsCache, index: int) -> torch.Tensor:
            ipt = cache.extract(index, eval_kwargs)
            y = eval_module(*ipt.args, **ipt.kwargs)
            y = y[0] if not isinstance(y, torch.Tensor) else y
            assert isinstance(y, torch.Tensor), "eval_mod should return a tensor"
            return y.contiguous().view(-1, y.shape[-1])

        saved = []
        if orig_x_wgts is not None:
            saved = [(p, p.data) for p, _ in orig_x_wgts]
            for p, ow in orig_x_wgts:
                p.data = ow.to(device=p.data.device)
        ref = orig_ipts or ipts
        base = [forward(ref, i) for i in range(len(ref.front().data))]
        for p, ow in saved:
            p.data = ow

        def score(qw: torch.Tensor) -> torch.Tensor:
            kept = param.data
            param.data = qw
            try:
                rows = None
                for i, y0 in enumerate(base):
                    y = forward(ipts, i)
                    rows = engine.product_error(y, y0, degree) if rows is None else \
                        torch.cat([rows, engine.product_error(y, y0, degree)])
            finally:
                param.data = kept
            return rows.double().sum().to(torch.float32).view(1)

        return score

    def _tell_general(self, error: torch.Tensor, candidate: DynamicRange, ratio: float, device) -> None:
        """calib/range.py:183-221.  Per channel group the reference's own update cannot run (it reshapes the mask
        expanded over rows to the per-channel-group ratio shape, calib/range.py:208-215, which raises); the mask is
        applied to the ratio before it is expanded here."""
        clamp_based = candidate.max is not None
        if self.best_error is None:
            self.best_error = error.clone()
            if clamp_based:
                self.best_range.min = None if candidate.min is None else candidate.min.expand(self.range_shape).clone()
                self.best_range.max = candidate.max.expand(self.range_shape).clone()
            self.best_range.ratio = torch.full(size=self.ratio_shape, fill_value=ratio, device=device,
                                               dtype=self.develop_dtype)
            return
        # (masked selections on the device: indexing with a boolean mask would drain the stream for every candidate)
        pos = error < self.best_error  # strict: the earlier (larger) ratio wins ties
        self.best_error = torch.where(pos, error, self.best_error)
        self.best_range.ratio.masked_fill_(pos.view(self.ratio_shape), ratio)
        if not clamp_based:
            return
        if self.granularity == SearchBasedCalibGranularity.Group:
            rpos = pos.view(self.range_shape)
        elif self.granularity == SearchBasedCalibGranularity.ChannelGroup:
            rpos = pos.view(self.pos_view_shape).expand(*self.range_shape)
        else:
            rpos = pos.view([1] * len(self.range_shape)).expand(*self.range_shape)
        self.best_range.max = torch.where(rpos, candidate.max.expand(self.range_shape), self.best_range.max)
        if isinstance(candidate.min, torch.Tensor):
            self.best_range.min = torch.where(rpos, candidate.min.expand(self.range_shape), self.best_range.min)

    def get_best(self) -> DynamicRange:
        """calib/range.py:152-164."""
        if self.static:
            return DynamicRange(min=self.best_range.min, max=self.best_range.max)
        if self.is_clamp_based():
            return DynamicRange(min=self.best_range.min, max=self.best_range.max, ratio=1.0)
        return DynamicRange(ratio=self.best_range.ratio.view(self.ratio_view_shape))

    # ---- activations (calib/range.py:101-150, calib/search.py:839-1078) --------------------------------------
    def _calibrate_acts(self, acts, mods, eval_inputs, eval_module, eval_kwargs, orig_wgts, orig_eval_inputs,
                        record_errors: bool, wgts=None, orig_acts=None) -> None:
        """Range of a module's inputs or outputs.  The cached batches, flattened ahead of the channel dimension, are
        viewed with the whole first dimension inside the group (``skip_first_dim``), so a static range is shared by
        all tokens.  Clamp-based search (static tensors, or ``allow_scale`` off): candidates are the measured range
        times each ratio; otherwise the candidate is the ratio itself, applied to the range the quantizer measures
        on the fly.  A candidate is scored on the tensor (inputs only: |Q(x) - x|^degree per channel group or in
        total, summed per batch in the tensor's dtype and added up in that dtype as the reference does) or on the
        evaluation module's outputs with the quantizer hooked onto the modules, or (inputs only) on the products of
        the quantized input with the weights of the linears that consume it."""
        what = "input" if self.tensor_type == TensorType.Inputs else "output"
        assert acts is not None, f"The {what} activations should be provided."
        assert acts.num_tensors == 1, f"Only one {what} is allowed, got {acts.num_tensors}"
        cache = acts.front()
        q, zd = self.quantizer, q_zero_domain(self.quantizer)
        is_fp = q.config.quant_dtype.is_float_point
        # objective / granularity as the reference resolves them (calib/search.py:361-414)
        self.objective, self.granularity = self.config.objective, self.config.granularity
        if self.tensor_type == TensorType.Outputs:
            self.objective, self.granularity = SearchBasedCalibObjective.OutputsError, SearchBasedCalibGranularity.Layer
        elif self.objective == SearchBasedCalibObjective.OutputsError:
            assert eval_inputs is not None, "eval_inputs should not be None when objective is OutputsError"
            assert eval_module is not None, "eval_module should not be None when OutputsError"
            if (isinstance(eval_module, (nn.Linear, nn.Conv2d))
                    and self.granularity.value < SearchBasedCalibGranularity.Layer.value):
                self.objective = SearchBasedCalibObjective.ProductsError
            else:
                self.granularity = SearchBasedCalibGranularity.Layer
        if self.pre_scale is not None:
            raise self._unsupported("progressive quantizers on activations")
        device = _device_of(cache, mods, eval_module)
        batches = [t.to(device) for t in cache.get_standardized_data(reshape=False)]
        assert all(t.shape == batches[0].shape for t in batches), "all batches must have the same shape"
        view = infer_view_shape(batches[0].shape, q.config.largest_group_shape, skip_first_dim=True)
        self._set_shapes_from_view(view)
        if self.is_clamp_based():
            self.base_range = DynamicRange.construct([t.view(view) for t in batches], zero_domain=zd,
                                                     is_float_point=is_fp)
        if self.objective == SearchBasedCalibObjective.TensorError:
            if self.granularity == SearchBasedCalibGranularity.Group:
                # calib/search.py:933-942: the Group branch falls through into the final else
                raise ValueError(f"Unknown granularity {self.granularity}")
            scorer = self._acts_tensor_scorer(cache, batches, view)
        elif self.objective == SearchBasedCalibObjective.ProductsError:
            assert self.tensor_type == TensorType.Inputs, "ProductsError scores inputs (or weights)"
            assert wgts, "wgts should not be None when objective is ProductsError"
            scorer = self._acts_products_scorer(acts, wgts, orig_wgts, orig_acts or orig_eval_inputs, device)
        else:
            assert self.objective == SearchBasedCalibObjective.OutputsError
            scorer = self._acts_outputs_scorer(mods, eval_inputs, eval_module, eval_kwargs, orig_wgts, orig_eval_inputs)
        self.best_range, self.best_error = DynamicRange(), None
        history = []
        for iteration in self.ratios:
            for ratio in iteration:
                if self.is_clamp_based():
                    candidate = self.base_range.scale(ratio=ratio, zero_domain=zd, is_float_point=is_fp)
                else:
                    candidate = DynamicRange(ratio=ratio)
                error = scorer(candidate)
                if record_errors:
                    history.append(error.clone())
                self._tell_general(error, candidate, ratio, device)
        if record_errors:
            self.candidate_errors = torch.stack(history)

    def _acts_tensor_scorer(self, cache, batches: list[torch.Tensor], view: torch.Size):
        """calib/search.py:928-946.  One launch of the error kernel per batch gives the float32 sums of every
        (row, group); they are added up over the rows (and groups) in float64 and rounded once to the tensor's
        dtype, where the reference's single multi-axis ``sum`` rounds once as well."""
        fine = torch.Size([batches[0].shape[0], 1, *view[2:]])  # one row per group: the quantizer's own groups
        degree = float(self.config.degree)
        refs = [engine.canonical_groups(t, fine)[0] for t in batches]
        channel_group = self.granularity == SearchBasedCalibGranularity.ChannelGroup

        def score(candidate: DynamicRange) -> torch.Tensor:
            total = None
            for x, b2d in zip(batches, refs):
                qx = self.quantizer.quantize(x, kernel=None, channels_dim=None, dynamic_range=candidate,
                                             default_dtype=x.dtype, develop_dtype=self.develop_dtype).data
                a2d, _ = engine.canonical_groups(qx, fine)
                raw = engine.group_error(a2d, b2d, degree).view(fine[::2]).double()
                if channel_group:
                    e = raw.sum(dim=[d for d in range(raw.ndim) if d != 1]).view(self.error_shape)
                else:
                    e = raw.sum().view(1)
                e = e.to(x.dtype)
                total = e if total is None else total.add_(e)
            return total

        return score

    def _acts_products_scorer(self, acts: TensorsCache, wgts, orig_wgts, orig_acts: TensorsCache | None, device):
        """calib/search.py:871-901, 947-973.  Every cached batch x[n, C] is multiplied with every weight w[R, C] of
        the linears that consume it, one quantization group of input channels at a time (a batched GEMM over the
        C/g groups; the whole product for Layer granularity), in the tensors' dtype; the candidate's error is
        sum float32(dtype(Q(x) w - x0 w0))^degree per channel group, added up over batches and weights in float32.
        The weights are not processed (the reference quantizes only the tensor being calibrated)."""
        view = self.view_shape
        if len(view) != 4:
            raise self._unsupported("ProductsError for inputs that are not [tokens, channels]")
        CG, g = view[2], view[3]
        layer = self.granularity == SearchBasedCalibGranularity.Layer
        degree = float(self.config.degree)

        def parse(cache: TensorsCache | None):
            if cache is None:
                return None
            return cache.front().repartition(max_batch_size=self.config.element_batch_size,
                                             max_size=self.config.element_size, standardize=True, reshape=False)

        ipt, orig_ipt = parse(acts), parse(orig_acts)
        assert ipt.channels_dim == 1
        batches = [x.to(device).view(x.shape[0], -1) for x in ipt.data]
        orig_batches = batches if orig_ipt is None else [x.to(device).view(x.shape[0], -1) for x in orig_ipt.data]
        weights = [w.detach().data.to(device) for w in wgts]
        originals = weights if not orig_wgts else [w.to(device) for _, w in orig_wgts]
        assert len(originals) == len(weights)
        assert all(w.ndim == 2 for w in weights), "ProductsError for inputs takes the weights of linear modules"

        def k_major(w: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
            """[R, C] -> [groups, g, R] in the inputs' dtype: the right-hand operand of x[:, group] @ w[:, group]^T."""
            w = w.to(like.dtype)
            if layer:
                return w.t().contiguous().view(1, w.shape[1], w.shape[0])
            return w.view(w.shape[0], CG, g).permute(1, 2, 0).contiguous()

        def products(x: torch.Tensor, wk: torch.Tensor) -> torch.Tensor:  # float64 accumulation, rounded once
            return engine.grouped_products(x, wk, x.dtype).view(1 if layer else CG, -1)

        weights = [k_major(w, batches[0]) for w in weights]
        originals = [k_major(w, orig_batches[0]) for w in originals]
        base = [[products(x, w) for x in orig_batches] for w in originals]

        def score(candidate: DynamicRange) -> torch.Tensor:
            total = None
            processed = [self.quantizer.quantize(x, kernel=None, channels_dim=None, dynamic_range=candidate,
                                                 default_dtype=x.dtype, develop_dtype=self.develop_dtype).data
                         for x in batches]
            for w, y0s in zip(weights, base):
                for x, y0 in zip(processed, y0s, strict=True):
                    total = engine.product_error(products(x, w), y0, degree, out=total)
            return total.view(self.error_shape)

        return score

    def _acts_outputs_scorer(self, mods, eval_inputs: TensorsCache, eval_module: nn.Module, eval_kwargs: dict,
                             orig_wgts, orig_eval_inputs):
        """calib/search.py:902-918, 974-989 (inputs) and :1010-1073 (outputs): the evaluation module's outputs with
        the candidate quantizer hooked onto ``mods`` (ahead of them for inputs, behind them for outputs) against its
        outputs with nothing quantized (and the original weights in place); one float32 scalar per candidate."""
        assert eval_inputs is not None, "eval_inputs should not be None when objective is OutputsError"
        assert eval_module is not None and mods, "eval_module and the modules to hook are required"
        ipts, orig_ipts = self._parse_ipts(eval_inputs), self._parse_ipts(orig_eval_inputs)
        degree = float(self.config.degree)
        is_output = self.tensor_type == TensorType.Outputs

        def forward(cache: TensorsCache, index: int) -> torch.Tensor:
            ipt = cache.extract(index, eval_kwargs)
            y = eval_module(*ipt.args, **ipt.kwargs)
            y = y[0] if not isinstance(y, torch.Tensor) else y
            assert isinstance(y, torch.Tensor), "eval_mod should return a tensor"
            return y.contiguous().view(-1, y.shape[-1])

        saved = [(p, p.data) for p, _ in orig_wgts]
        for p, ow in orig_wgts:
            p.data = ow.to(device=p.data.device)
        ref = orig_ipts or ipts
        base = [forward(ref, i) for i in range(len(ref.front().data))]
        for p, ow in saved:
            p.data = ow

        def score(candidate: DynamicRange) -> torch.Tensor:
            def process(t: torch.Tensor) -> torch.Tensor:
                return self.quantizer.quantize(t, kernel=None, dynamic_range=candidate, default_dtype=t.dtype,
                                               develop_dtype=self.develop_dtype).data

            hook = self.quantizer.as_hook(process, is_output=is_output).register(mods)
            try:
                total = None
                for i, y0 in enumerate(base):
                    e = engine.product_error(forward(ipts, i), y0, degree).double().sum().to(torch.float32).view(1)
                    total = e if total is None else total.add_(e)
            finally:
                hook.remove()
            return total

        return score


def _device_of(cache, mods, eval_module) -> torch.device:
    """Where an activation search runs: the GPU the cached batches live on, else the one the modules live on."""
    for t in cache.data:
        if t.is_cuda:
            return t.device
    for m in list(mods or []) + ([eval_module] if eval_module is not None else []):
        for p in m.parameters():
            if p.is_cuda:
                return p.device
    dev = torch.device(cache.orig_device)
    return dev if dev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())


def q_zero_domain(quantizer: Quantizer):
    return quantizer.config.zero_domain


def _first_strict_min(errors: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """errors[K, ...] -> (smallest error, index of the first candidate reaching it): what the reference's
    'replace when strictly smaller' update keeps."""
    best = errors.min(dim=0).values
    first = (errors == best.unsqueeze(0)).to(torch.uint8).argmax(dim=0)
    return best, first


def calibrate_dynamic_range(tensor_type: TensorType, config: DynamicRangeCalibConfig | None, static: bool,
                            quantizer: Quantizer, modules: tp.Sequence[nn.Module], activations: TensorsCache,
                            weights: tp.Sequence[nn.Parameter] | None = None, eval_inputs: TensorsCache | None = None,
                            eval_module: nn.Module | None = None, eval_kwargs: dict[str, tp.Any] | None = None,
                            orig_weights=None, orig_activations=None, orig_eval_inputs=None, *, reduce_errors=None
                            ) -> tp.Sequence[DynamicRange] | None:
    """Per-step dynamic ranges of ``quantizer`` (None when nothing has to be stored); calib/range.py:302-474.
    ``reduce_errors``: for a weight whose rows are sharded over processes, the sum over shards of a float64 tensor of
    partial errors (see ``deepcompressor_amd.parallel.search_range_row_sharded``)."""
    if config is None or not quantizer.is_enabled():
        return None
    decomposed = quantizer.config.decompose()
    num_steps = decomposed.num_steps
    if not config.needs_search and (not static or tensor_type == TensorType.Weights):
        if config.ratio != 1.0:
            return tuple([DynamicRange(ratio=config.ratio)] + [None] * (num_steps - 1))
        return None
    if weights is None:
        weights = [m.weight for m in modules if hasattr(m, "weight")]
    if tensor_type == TensorType.Weights:
        assert len(modules) == 1 and len(weights) == 1, "only one module / weight per weight-range calibration"
        if eval_module is None:
            eval_module = modules[0]
            if eval_inputs is None:
                eval_inputs = activations
        else:
            assert eval_inputs is not None, "eval_inputs is required when eval_module is provided"
    else:
        assert activations is not None, "activations is required for activation quantization calibration"
        assert activations.num_tensors == 1, "only one tensor is supported for activation quantization calibration"
    if tensor_type != TensorType.Outputs:
        search_args = dict(x_acts=activations, x_mods=list(modules), eval_inputs=eval_inputs, eval_module=eval_module,
                           eval_kwargs=eval_kwargs, orig_x_wgts=orig_weights, orig_x_acts=orig_activations,
                           orig_eval_inputs=orig_eval_inputs)
        x_wgts = list(weights)
    else:
        search_args = dict(y_wgts=list(weights), y_acts=activations, y_mods=list(modules), eval_inputs=eval_inputs,
                           eval_module=eval_module, eval_kwargs=eval_kwargs, orig_y_wgts=orig_weights,
                           orig_y_acts=orig_activations, orig_eval_inputs=orig_eval_inputs)
        x_wgts = []
    if num_steps == 1:
        calibrator = DynamicRangeCalibrator(tensor_type=tensor_type, config=config, static=static, quantizer=quantizer,
                                            reduce_errors=reduce_errors)
        return (calibrator.calibrate(x_wgts=x_wgts, **search_args),)
    # progressive quantizer: one calibrator per step, each seeing the tensor divided by the scales found so far
    if tensor_type != TensorType.Weights:
        raise NotImplementedError("progressive range search is implemented for weights only")
    tensor = weights[0].detach().data
    develop_dtype = quantizer.develop_dtype
    default_scale_dtype = quantizer.default_dtype or tensor.dtype
    develop_tensor = tensor.to(dtype=develop_dtype) if tensor.dtype != develop_dtype else tensor.clone()
    info = QuantInfo.construct(decomposed, tensor_shape=develop_tensor.shape, default_dtype=default_scale_dtype,
                               quant_range=quantizer.quant_range, range_bound=quantizer.range_bound)
    dynamic_ranges, quant_scale = [], QuantScale()
    for step, step_info in enumerate(info.steps):
        step_quantizer = Quantizer(config=step_info.to_config(), kernel=quantizer.kernel if step == num_steps - 1 else None,
                                   quant_range=step_info.quant_range, range_bound=step_info.range_bound,
                                   default_dtype=quantizer.default_dtype, develop_dtype=quantizer.develop_dtype)
        step_range = DynamicRangeCalibrator(tensor_type=tensor_type, config=config, static=static,
                                            quantizer=step_quantizer, pre_scale=quant_scale.data,
                                            reduce_errors=reduce_errors).calibrate(x_wgts=list(weights), **search_args)
        dynamic_ranges.append(step_range)
        step_scale, _ = step_info.scale.quantize(tensor=develop_tensor.view(step_info.tensor_shape),
                                                 dynamic_range=step_range)
        quant_scale.append(step_scale)
        if num_steps > 2 and step < num_steps - 1:
            step_quant_range = step_info.tensor_quant_range
            develop_tensor = develop_tensor.view(step_info.tensor_view_shape).div_(step_scale.data)
            develop_tensor = develop_tensor.clamp_(min=step_quant_range.min, max=step_quant_range.max)
    return tuple(dynamic_ranges)
