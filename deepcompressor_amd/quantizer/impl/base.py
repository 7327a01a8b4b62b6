"""Fake-quantization driver (reference: deepcompressor/quantizer/impl/base.py:22-323).

``QuantizerImpl.quantize`` has the reference's signature and results.  Two execution paths, both HIP:

* fused   -- one step, one scale level, round-to-nearest: a single kernel reads the tensor once and writes
             the dequantized tensor (and / or codes) once;
* general -- multi-level scales, two progressive steps, explicit scales or a caller-supplied kernel object:
             range kernel -> scale-tree kernels -> quantize / dequantize kernel.
"""

from __future__ import annotations

import typing as tp
from dataclasses import dataclass, field

import torch

from ... import engine
from ...data.range import DynamicRange, QuantRange, RangeBound
from ...data.scale import QuantScale
from ...data.tensor import QuantTensor
from ...data.zero import ZeroPointDomain
from ...utils.config import KeyEnableConfig
from ..config.base import BaseQuantizerConfig
from ..config.kernel import BaseQuantKernel, BaseQuantKernelConfig
from ..kernel.gptq import QuantGptqKernel
from ..kernel.rtn import QuantRtnKernel
from .info import QuantInfo

__all__ = ["QuantizerImpl"]


def _per_group(values: torch.Tensor, view_shape, n: int) -> torch.Tensor:
    """A (#g0, 1, #g1, 1, ...) tensor (or a coarser one) as one contiguous value per group."""
    keep = [v if i % 2 == 0 else 1 for i, v in enumerate(view_shape)]
    if values.numel() == n:
        return values.reshape(n).contiguous()
    split = []
    for d in range(0, values.ndim, 2):
        split += [values.shape[d], keep[d] // values.shape[d]]
    return values.view([v if i % 2 == 0 else 1 for i, v in enumerate(split)]).expand(split).reshape(n).contiguous()


@dataclass
class QuantizerImpl:
    config: BaseQuantizerConfig | None
    key: str = ""
    info: QuantInfo | None = field(init=False, default=None)

    def is_enabled(self) -> bool:
        if self.config is None:
            return False
        if isinstance(self.config, KeyEnableConfig):
            return self.config.is_enabled_for(self.key)
        return self.config.is_enabled()

    def quantize(self, tensor: torch.Tensor, *, kernel: BaseQuantKernel | BaseQuantKernelConfig | None = None,
                 channels_dim: int | None = None, scale=None, zero: torch.Tensor | None = None, dynamic_range=None,
                 range_bound: RangeBound | None = None, quant_range: QuantRange | None = None,
                 return_with_dequant: bool = True, return_with_quant: bool = False,
                 default_dtype: torch.dtype | None = torch.float16, develop_dtype: torch.dtype = torch.float32,
                 **kwargs) -> QuantTensor:
        shape = tensor.shape
        if channels_dim is not None:
            tensor = tensor.reshape(-1, *shape[channels_dim:])
        round_delta = kwargs.pop("round_delta", None)
        if round_delta is not None:
            round_delta = round_delta.view(-1, *shape[channels_dim:])
        result = self._quantize(tensor, kernel=kernel, scale=scale, zero=zero, dynamic_range=dynamic_range,
                                range_bound=range_bound, quant_range=quant_range, round_delta=round_delta,
                                return_with_dequant=return_with_dequant, return_with_quant=return_with_quant,
                                default_dtype=default_dtype or tensor.dtype, develop_dtype=develop_dtype, **kwargs)
        if result.data is not None:
            result._dequantized = result.data.view(shape)
        if result.qdata is not None:
            result._quantized = result.qdata.view(shape)
        return result

    def _quantize(self, tensor: torch.Tensor, *, kernel=None, scale=None, zero=None, dynamic_range=None,
                  range_bound=None, quant_range=None, round_delta=None, return_with_dequant: bool = True,
                  return_with_quant: bool = False, default_dtype: torch.dtype = torch.float16,
                  develop_dtype: torch.dtype = torch.float32, **kwargs) -> QuantTensor:
        shape, dtype = tensor.shape, tensor.dtype
        self.update(shape, default_dtype, quant_range, range_bound)
        if self.info is None or self.info.num_steps == 0:
            return QuantTensor(dequantized=tensor, quantized=tensor, view_shape=shape)
        if develop_dtype != torch.float32:
            raise NotImplementedError("the HIP kernels compute in float32 (develop_dtype=torch.float32)")
        n_steps = self.info.num_steps
        # ---- normalise the per-step arguments
        if scale is None:
            scale = (None,) * n_steps
        elif not isinstance(scale, tp.Sequence):
            scale = (scale,)
        if dynamic_range is None:
            dynamic_range = (None,) * n_steps
        elif isinstance(dynamic_range, DynamicRange):
            dynamic_range = (dynamic_range,) if dynamic_range.is_set() else (None,) * n_steps
        assert isinstance(scale, (tuple, list)) and len(scale) == n_steps, "scale must have one entry per step"
        assert isinstance(dynamic_range, (tuple, list)) and len(dynamic_range) == n_steps, (
            "dynamic_range must have one entry per step")
        if isinstance(kernel, BaseQuantKernelConfig):
            kernel = kernel.build()
        native_rtn = kernel is None or type(kernel) is QuantRtnKernel
        if not native_rtn:  # GPTQ / caller-supplied kernel objects: their codes are checked where they are made
            return self._dispatch(tensor, kernel, scale, zero, dynamic_range, round_delta, return_with_dequant,
                                  return_with_quant, kwargs)
        # The kernels report a NaN / Inf in what they read into the thread's flag word; see _refuse_nonfinite
        with engine.finite_scope(tensor.device):
            result = self._dispatch(tensor, kernel, scale, zero, dynamic_range, round_delta, return_with_dequant,
                                    return_with_quant, kwargs)
        self._refuse_nonfinite(tensor, scale, dynamic_range, return_with_quant)
        return result

    def _dispatch(self, tensor, kernel, scale, zero, dynamic_range, round_delta, return_with_dequant,
                  return_with_quant, kwargs) -> QuantTensor:
        n_steps = self.info.num_steps
        if n_steps > 2:
            return self._quantize_chain(tensor, kernel, scale, zero, dynamic_range, round_delta, return_with_dequant,
                                        return_with_quant, kwargs)
        native_rtn = kernel is None or type(kernel) is QuantRtnKernel
        last = self.info.steps[-1]
        dr_last = dynamic_range[-1]
        fp_with_zero = last.scale.has_zero_point and last.quant_dtype.is_float_point  # range centred on the mean
        fusable = (n_steps == 1 and native_rtn and round_delta is None and scale[0] is None
                   and last.scale.is_single_level() and not fp_with_zero
                   and not (dr_last is not None and dr_last.ratio is not None
                            and (dr_last.max is not None or isinstance(dr_last.ratio, torch.Tensor))))
        if fusable:
            return self._quantize_fused(tensor, dr_last, return_with_dequant, return_with_quant)
        ratio_only = dr_last is None or (dr_last.max is None and not isinstance(dr_last.ratio, torch.Tensor))
        if (n_steps == 1 and native_rtn and round_delta is None and scale[0] is None and ratio_only
                and last.scale.is_two_level_per_tensor() and engine.groups_are_contiguous(last.tensor_view_shape)):
            return self._quantize_two_level(tensor, dr_last, return_with_dequant, return_with_quant)
        return self._quantize_general(tensor, kernel, scale, zero, dynamic_range, round_delta, return_with_dequant,
                                      return_with_quant, kwargs)

    @staticmethod
    def _refuse_nonfinite(tensor, scale, dynamic_range, want_quant: bool) -> None:
        """The reference asserts that the codes hold neither NaN nor Inf (impl/base.py:250-251).  The round-to-nearest
        kernels here clamp with min / max instructions, which drop a NaN operand, so their codes cannot show it; the
        condition is caught where it arises -- a NaN anywhere in the input gives the reference NaN codes, and so does
        an Inf when the step's range is taken from the data (Inf / Inf) -- by the kernels themselves while they read
        the tensor (no extra pass).  This is the host's look at their flag: one blocking 4-byte read, by default
        before every return as the reference does (``engine.FINITE_CHECKS``: "always"; "export": only for calls that
        hand back codes; "deferred": at ``engine.raise_if_nonfinite()``; "never")."""
        policy = engine.FINITE_CHECKS
        if policy in ("never", "deferred") or (policy == "export" and not want_quant):
            return
        if not engine.take_nonfinite(tensor.device):
            return
        # rare path: tell NaN from Inf as the reference's two assertions do
        if policy != "always" and bool(torch.isfinite(tensor).all()):
            # not this tensor: the flag was left by an earlier call of this thread that nobody has looked at yet
            raise AssertionError("Quantized tensor contains NaN or Inf (a tensor quantized earlier on this thread, "
                                 "whose check was deferred).")
        assert not tensor.isnan().any(), "Quantized tensor contains NaN."
        from_data = all(s is None for s in scale) and all(
            d is None or (d.max is None and not isinstance(d.ratio, torch.Tensor)) for d in dynamic_range)
        if from_data and bool(tensor.isinf().any()):
            raise AssertionError("Quantized tensor contains NaN.")  # the reference's Inf / Inf

    # ---- fused path --------------------------------------------------------------------------
    def _quantize_fused(self, tensor, dyn, want_dequant, want_quant) -> QuantTensor:
        step_info = self.info.steps[-1]
        sinfo = step_info.scale
        view = step_info.tensor_view_shape
        keep = [v if i % 2 == 0 else 1 for i, v in enumerate(view)]
        step = sinfo.native(tensor.device)
        x2d, restore = engine.canonical_groups(tensor, view)
        rmax = rmin = ratio = None
        if dyn is not None:
            if dyn.max is not None:
                fmt = lambda v: DynamicRange._format_m_(  # noqa: E731
                    v, shape=torch.Size(keep), dtype=torch.float32, device=tensor.device).contiguous().view(-1)
                rmax = fmt(dyn.max)
                rmin = fmt(dyn.min) if sinfo.has_zero_point else None
            else:
                ratio = dyn.ratio
        both = want_dequant and want_quant
        # with both outputs requested the reference hands back the *dequantized float32 values* as qdata
        # (its qdata aliases the buffer it dequantizes in place); reproduce that, see QuantTensor docs
        dq, codes, s, z = engine.fake_quant_fused(step, x2d, range_max=rmax, range_min=rmin, ratio=ratio,
                                                  want_dequant=want_dequant, want_codes=want_quant,
                                                  want_scales=want_quant)
        scale = QuantScale().append(s.view(keep)) if want_quant else None
        zero = None
        if want_quant:  # (a host-to-device copy when there is no zero point: only when the caller wants it)
            zero = z.view(keep) if sinfo.has_zero_point else torch.tensor(0, dtype=torch.float32, device=tensor.device)
        quantized = None
        if want_quant:
            if both:
                codes = engine.dequantize_groups(step, codes, s, z if sinfo.has_zero_point else None)
            quantized = restore(codes)
        return QuantTensor(dequantized=restore(dq) if want_dequant else None, quantized=quantized,
                           scale=QuantScale().append(scale) if want_quant else None,
                           zero=zero, view_shape=view if want_quant else None)

    # ---- two scale levels, the first per tensor (NVFP4) ----------------------------------------
    def _quantize_two_level(self, tensor, dyn, want_dequant, want_quant) -> QuantTensor:
        """One reduction (the tensor's largest magnitude) and one pass, instead of range -> two scale-level kernels
        -> join -> quantize over separate per-group arrays."""
        step_info = self.info.steps[-1]
        sinfo = step_info.scale
        view = step_info.tensor_view_shape
        keep = [v if i % 2 == 0 else 1 for i, v in enumerate(view)]
        device = tensor.device
        step = sinfo.native(device)
        level0 = engine.make_rounding(sinfo.linear_scale_quant_dtypes[0], has_zero_point=False, quant_range=None,
                                      device=device)
        x2d, restore = engine.canonical_groups(tensor, view)
        ratio = None if dyn is None else dyn.ratio
        dq, codes, first, second, compound = engine.fake_quant_two_level(
            step, level0, sinfo.linear_scale_quant_spans[0], x2d, ratio=ratio, want_dequant=want_dequant,
            want_codes=want_quant, want_scales=want_quant)
        quantized = scale = zero = None
        if want_quant:
            scale = QuantScale().append(first.view([1] * len(view))).append(second.view(keep))
            scale.data = compound.view(keep)  # (a zero compound is 1: data/scale.py:66-69)
            zero = torch.tensor(0, dtype=torch.float32, device=device)
            if want_dequant:  # the reference's qdata aliases its in-place float32 dequantization buffer
                codes = engine.dequantize_groups(step, codes, compound, None)
            quantized = restore(codes)
        return QuantTensor(dequantized=restore(dq) if want_dequant else None, quantized=quantized,
                           scale=QuantScale().append(scale) if want_quant else None, zero=zero,
                           view_shape=view if want_quant else None)

    # ---- more than two progressive steps ------------------------------------------------------
    def _quantize_chain(self, tensor, kernel, scale, zero, dynamic_range, round_delta, want_dequant, want_quant,
                        kwargs) -> QuantTensor:
        """Three or more progressive steps (the reference's loop takes any number, impl/base.py:211-279; none of its
        example configurations has more than two).  Every step but the last two is applied on its own -- the step's
        scale from the developed tensor, then develop <- clamp(develop / scale) in float32 --, the last two run as a
        two-step quantizer on the developed tensor, and the de-quantization walks back through the earlier steps,
        clamping to each step's dtype on the way (base.py:270-278)."""
        info, shape, dtype = self.info, tensor.shape, tensor.dtype
        n_steps = info.num_steps
        develop = tensor.to(torch.float32)
        earlier: list[QuantScale] = []
        for i in range(n_steps - 2):
            step_info = info.steps[i]
            step_scale, _ = step_info.scale.quantize(scale=scale[i], zero=None, tensor=develop,
                                                     dynamic_range=dynamic_range[i], outer=None, outer_range=None)
            earlier.append(step_scale)
            bounds = step_info.tensor_quant_range
            develop = develop.view(step_info.tensor_view_shape).div(step_scale.data).view(shape)
            develop = develop.clamp_(min=bounds.min, max=bounds.max)
        # (the shortened plan is handed down, not swapped into ``self.info``: host threads that share this quantizer,
        # one per HIP stream, must never see it)
        last_two = QuantInfo(steps=info.steps[-2:], needs_dequant_saturation=info.needs_dequant_saturation)
        inner = self._quantize_general(develop, kernel, tuple(scale[-2:]), zero, tuple(dynamic_range[-2:]),
                                       round_delta, want_dequant, want_quant, kwargs, info=last_two)
        dequantized = None
        if want_dequant:
            d = inner.data
            for i in range(n_steps - 3, -1, -1):
                step_info = info.steps[i]
                d = d.clamp_(min=step_info.quant_dtype.min_value, max=step_info.quant_dtype.max_value)
                d = d.view(step_info.tensor_view_shape).mul_(earlier[i].data).view(shape)
            dequantized = d.to(dtype)
        quantized = quant_scale = None
        if want_quant:
            # with both outputs the reference's qdata aliases the float32 buffer it de-quantizes in place
            quantized = d if want_dequant else inner.qdata
            quant_scale = QuantScale()
            for step_scale in earlier:
                quant_scale.append(step_scale)
            for k in range(2):
                quant_scale.append(inner.scale.get_child(k))
        return QuantTensor(dequantized=dequantized, quantized=quantized, scale=quant_scale,
                           zero=inner.zero if want_quant else None, view_shape=inner.view_shape if want_quant else None)

    # ---- general path ------------------------------------------------------------------------
    def _quantize_general(self, tensor, kernel, scale, zero, dynamic_range, round_delta, want_dequant, want_quant,
                          kwargs, info: QuantInfo | None = None) -> QuantTensor:
        info = self.info if info is None else info
        shape, dtype, device = tensor.shape, tensor.dtype, tensor.device
        n_steps = info.num_steps
        quant_scale = QuantScale()
        outer = outer_range = None  # previous step: per-group divisor (in the current step's group order) and clamp
        outer_scale_data = None
        native_rtn = kernel is None or type(kernel) is QuantRtnKernel
        if n_steps == 2 and native_rtn and round_delta is None and zero is None and scale[0] is None \
                and scale[1] is None:
            fused = self._quantize_two_steps_fused(tensor, dynamic_range, want_dequant, want_quant, info)
            if fused is not None:
                return fused
        for i, step_info in enumerate(info.steps):
            sinfo = step_info.scale
            n_groups = sinfo.num_groups
            if outer_scale_data is not None:
                outer = _per_group(outer_scale_data, step_info.tensor_view_shape, n_groups)
            if i == n_steps - 1 and outer is not None and native_rtn and round_delta is None and scale[i] is None \
                    and zero is None:
                fused = self._quantize_last_step_fused(tensor, step_info, dynamic_range[i], outer, outer_range,
                                                       quant_scale, want_dequant, want_quant, info)
                if fused is not None:
                    return fused
            step_scale, step_zero = sinfo.quantize(scale=scale[i], zero=None if i < n_steps - 1 else zero,
                                                   tensor=tensor, dynamic_range=dynamic_range[i], outer=outer,
                                                   outer_range=outer_range)
            quant_scale.append(step_scale)
            if i < n_steps - 1:
                outer_scale_data, outer_range = step_scale.data, step_info.tensor_quant_range
        last = info.steps[-1]
        sinfo, view = last.scale, last.tensor_view_shape
        step = sinfo.native(device)
        x2d, restore = engine.canonical_groups(tensor, view)
        s = step_scale.data.contiguous().view(-1)
        z = step_zero.contiguous().view(-1) if sinfo.has_zero_point else None
        lo, hi = (outer_range.min, outer_range.max) if outer is not None else (0.0, 0.0)
        clamp = False
        deq_lo = deq_hi = 0.0
        if outer is not None:
            first = info.steps[0]
            clamp = info.needs_dequant_saturation
            deq_lo, deq_hi = first.quant_dtype.min_value, first.quant_dtype.max_value
        native_rtn = kernel is None or type(kernel) is QuantRtnKernel
        both = want_dequant and want_quant
        dq = codes = None
        if native_rtn:
            delta = None
            if round_delta is not None:  # impl/base.py:240-247: one delta row may serve every row group
                assert not last.quant_dtype.is_float_point, "round_delta is not supported for float quantization"
                rd = round_delta.view(1, 1, *view[2:]) if round_delta.shape[0] == 1 else round_delta.view(view)
                delta, _ = engine.canonical_groups(rd.to(torch.float32).expand(view).contiguous(), view)
            dq, codes = engine.quantize_groups(step, x2d, s, z, outer=outer, outer_lo=lo, outer_hi=hi, deq_clamp=clamp,
                                               deq_lo=deq_lo, deq_hi=deq_hi, out_dtype=dtype,
                                               want_dequant=want_dequant, want_codes=want_quant, round_delta=delta)
        else:
            # caller-supplied kernel object: it receives the float32 working tensor of the last step
            if outer is not None:
                develop = restore(engine.prescale_groups(x2d, outer, lo, hi))
            elif type(kernel) is QuantGptqKernel:
                develop = tensor  # that kernel widens and permutes the weight in one pass and leaves its input alone
            else:
                develop = tensor.to(torch.float32) if dtype != torch.float32 else tensor.clone()
            rd = None
            if round_delta is not None:
                rd = round_delta.view(1, 1, *view[2:]) if round_delta.shape[0] == 1 else round_delta.view(view)
            codes_view = kernel.quantize(tensor=develop, view_shape=view, quant_dtype=last.quant_dtype,
                                         zero_domain=last.zero_domain, scale=step_scale.data, zero=step_zero,
                                         quant_range=last.quant_range, range_bound=last.range_bound, round_delta=rd,
                                         **kwargs)
            if type(kernel) is not QuantGptqKernel:  # (that kernel has made the same check on the same values)
                engine.assert_finite(codes_view, "Quantized tensor")
            codes, _ = engine.canonical_groups(codes_view.reshape(shape), view)
            if want_dequant:
                dq = engine.dequantize_groups(step, codes, s, z, outer=outer, deq_clamp=clamp, deq_lo=deq_lo,
                                              deq_hi=deq_hi, out_dtype=dtype)
        quantized = None
        if want_quant:
            if both:  # the reference's qdata aliases its in-place float32 dequantization buffer
                codes = engine.dequantize_groups(step, codes, s, z, outer=outer, deq_clamp=clamp, deq_lo=deq_lo,
                                                 deq_hi=deq_hi, out_dtype=torch.float32)
            quantized = restore(codes)
        return QuantTensor(dequantized=restore(dq) if want_dequant else None, quantized=quantized,
                           scale=quant_scale if want_quant else None, zero=step_zero if want_quant else None,
                           view_shape=view if want_quant else None)

    def smooth_candidates(self, tensor: torch.Tensor, chan_scale: torch.Tensor, *, divide_first: bool,
                          channels_dim: int | None = None, scale=None, zero=None, dynamic_range=None,
                          range_bound: RangeBound | None = None, quant_range: QuantRange | None = None,
                          default_dtype: torch.dtype | None = None, errors_degree: float | None = None,
                          as_codes: bool = False) -> torch.Tensor | tuple[torch.Tensor, torch.Tensor] | None:
        """A smoothing candidate through the fused kernel (reference op chain: calib/smooth.py:504-530).

        ``chan_scale`` [C]: returns ``Q(tensor * s) / s`` (or ``Q(tensor / s) * s`` with ``divide_first``) in the
        tensor's dtype, equal bit for bit to scaling in float32, ``quantize`` and scaling back.  ``chan_scale``
        [K][C] with ``errors_degree``: returns the float32 [K][#groups] error sums |processed - tensor|^degree of all
        K candidates from one read of the tensor.  ``chan_scale`` [K][C] with ``as_codes``: returns the int8 codes
        [K][rows][C] (minus the zero point, when it lives in the code's domain) and float32 group scales
        [K][rows][#groups per row] whose product is the fake-quantized smoothed tensor of every candidate.  Returns
        None when the configuration is outside the fused kernel (several steps or scale levels, non-integer types,
        fixed scales or ranges, groups that are not runs of channels inside one row): the caller then takes the
        general path."""
        if not tensor.is_cuda or not tensor.is_contiguous():
            return None
        shape = tensor.shape
        if channels_dim is not None:
            tensor = tensor.reshape(-1, *shape[channels_dim:])
        if tensor.ndim != 2 or scale is not None or zero is not None:
            return None
        ratio = None
        if isinstance(dynamic_range, (tuple, list)):
            dynamic_range = dynamic_range[0] if len(dynamic_range) == 1 else "several"
        if dynamic_range is not None:
            if not isinstance(dynamic_range, DynamicRange):
                return None
            if dynamic_range.is_set():
                if dynamic_range.max is not None or isinstance(dynamic_range.ratio, torch.Tensor):
                    return None
                ratio = dynamic_range.ratio
        info = self.update(tensor.shape, default_dtype or tensor.dtype, quant_range, range_bound)
        if info is None or info.num_steps != 1:
            return None
        last = info.steps[-1]
        view = last.tensor_view_shape
        if not last.scale.is_single_level() or len(view) != 4 or view[1] != 1:
            return None
        if last.scale.has_zero_point and last.quant_dtype.is_float_point:
            return None
        step = last.scale.native(tensor.device)
        if not engine.smooth_step_is_fusable(step):
            return None
        rows, _, groups_per_row, group = view
        vec = 4 if tensor.dtype == torch.float32 else 8
        if group % vec or chan_scale.shape[-1] != groups_per_row * group or tensor.data_ptr() % 16:
            return None
        x2d = tensor.view(rows * groups_per_row, group)
        if as_codes:
            if not engine.smooth_codes_are_fusable(step):
                return None
            codes, scales = engine.smooth_quantize(step, x2d, groups_per_row, chan_scale, divide_first=divide_first,
                                                   ratio=ratio)
            return codes.view(-1, rows, groups_per_row * group), scales.view(-1, rows, groups_per_row)
        out = engine.smooth_fake_quant(step, x2d, groups_per_row, chan_scale, divide_first=divide_first, ratio=ratio,
                                       errors_degree=errors_degree)
        return out.view(shape) if errors_degree is None else out

    def _quantize_two_steps_fused(self, tensor, dynamic_range, want_dequant, want_quant, info) -> QuantTensor | None:
        """A two-step progressive quantizer whose first step has one group per row (QoQ W4: per-channel scales, then
        groups inside the row) in ONE pass over the tensor (``dcq_fake_quant_progressive``).  None when the
        configuration or the tensor is outside that kernel; the caller then goes step by step."""
        first, last = info.steps
        v0, v1 = first.tensor_view_shape, last.tensor_view_shape
        if tensor.ndim != 2 or not tensor.is_contiguous() or len(v0) != 4 or len(v1) != 4:
            return None
        if tensor.data_ptr() % 16:  # the kernel moves 16-byte pieces
            return None
        rows, row_len = tensor.shape
        if tuple(v0) != (rows, 1, 1, row_len) or v1[0] != rows or v1[1] != 1 or v1[2] * v1[3] != row_len:
            return None
        if not (first.scale.is_single_level() and last.scale.is_single_level()) or first.scale.has_zero_point:
            return None
        ratios = []
        for dyn in dynamic_range:
            if dyn is not None and dyn.is_set():
                if dyn.max is not None or isinstance(dyn.ratio, torch.Tensor):
                    return None
                ratios.append(dyn.ratio)
            else:
                ratios.append(None)
        device = tensor.device
        step0, step1 = first.scale.native(device), last.scale.native(device)
        group = v1[3]
        if not engine.progressive_is_fusable(step0, step1, row_len, group, tensor.dtype):
            return None
        outer_range = first.tensor_quant_range
        dq, codes, s0, s1, z1 = engine.fake_quant_progressive(
            step0, step1, tensor, group, outer_range.min, outer_range.max, deq_clamp=info.needs_dequant_saturation,
            deq_lo=first.quant_dtype.min_value, deq_hi=first.quant_dtype.max_value, first_ratio=ratios[0],
            ratio=ratios[1], want_dequant=want_dequant, want_codes=want_quant, want_scales=want_quant)
        quantized = zero = quant_scale = None
        if want_quant:
            keep0, keep1 = [rows, 1, 1, 1], [rows, 1, v1[2], 1]
            quant_scale = QuantScale()
            quant_scale.append(QuantScale().append(s0.view(keep0)))
            quant_scale.append(QuantScale().append(s1.view(keep1)))
            zero = z1.view(keep1) if last.scale.has_zero_point else torch.tensor(0, dtype=torch.float32, device=device)
            if want_dequant:  # the reference's qdata aliases its in-place float32 dequantization buffer
                outer = s0.view(rows, 1).expand(rows, v1[2]).reshape(-1)
                codes = engine.dequantize_groups(
                    step1, codes.view(-1, group), s1, z1 if last.scale.has_zero_point else None, outer=outer,
                    deq_clamp=info.needs_dequant_saturation, deq_lo=first.quant_dtype.min_value,
                    deq_hi=first.quant_dtype.max_value, out_dtype=torch.float32)
            quantized = codes.view(rows, row_len)
        return QuantTensor(dequantized=dq, quantized=quantized, scale=quant_scale, zero=zero,
                           view_shape=v1 if want_quant else None)

    def _quantize_last_step_fused(self, tensor, last, dyn, outer, outer_range, quant_scale, want_dequant,
                                  want_quant, info) -> QuantTensor | None:
        """The last step of a progressive quantizer in one pass over the tensor (``dcq_fake_quant_groups_outer``):
        range of clamp(x / outer), scale and zero of any one-level dtype, codes, dequantization back through the
        previous steps' scale.  None when the step needs the general path (several scale levels, a floating-point
        type with a zero point, a ratio on top of a fixed range)."""
        sinfo, view = last.scale, last.tensor_view_shape
        if not sinfo.is_single_level() or (sinfo.has_zero_point and last.quant_dtype.is_float_point):
            return None
        device, dtype = tensor.device, tensor.dtype
        keep = [v if i % 2 == 0 else 1 for i, v in enumerate(view)]
        rmax = rmin = ratio = None
        if dyn is not None and dyn.is_set():
            if dyn.max is not None:
                if dyn.ratio is not None:
                    return None
                fmt = lambda v: DynamicRange._format_m_(  # noqa: E731
                    v, shape=torch.Size(keep), dtype=torch.float32, device=device).contiguous().view(-1)
                rmax = fmt(dyn.max)
                rmin = fmt(dyn.min) if sinfo.has_zero_point else None
            elif isinstance(dyn.ratio, torch.Tensor):
                return None
            else:
                ratio = dyn.ratio
        step = sinfo.native(device)
        x2d, restore = engine.canonical_groups(tensor, view)
        first = info.steps[0]
        clamp = info.needs_dequant_saturation
        deq_lo, deq_hi = first.quant_dtype.min_value, first.quant_dtype.max_value
        both = want_dequant and want_quant
        dq, codes, s, z = engine.fake_quant_fused_outer(
            step, x2d, outer, outer_range.min, outer_range.max, deq_clamp=clamp, deq_lo=deq_lo, deq_hi=deq_hi,
            range_max=rmax, range_min=rmin, ratio=ratio, out_dtype=dtype, want_dequant=want_dequant,
            want_codes=want_quant, want_scales=want_quant)
        quantized = zero = None
        if want_quant:
            quant_scale.append(QuantScale().append(s.view(keep)))
            zero = z.view(keep) if sinfo.has_zero_point else torch.tensor(0, dtype=torch.float32, device=device)
            if both:  # the reference's qdata aliases its in-place float32 dequantization buffer
                codes = engine.dequantize_groups(step, codes, s, z if sinfo.has_zero_point else None, outer=outer,
                                                 deq_clamp=clamp, deq_lo=deq_lo, deq_hi=deq_hi, out_dtype=torch.float32)
            quantized = restore(codes)
        return QuantTensor(dequantized=restore(dq) if want_dequant else None, quantized=quantized,
                           scale=quant_scale if want_quant else None, zero=zero,
                           view_shape=view if want_quant else None)

    def update(self, tensor_shape: torch.Size, default_dtype: torch.dtype | None, quant_range: QuantRange | None,
               range_bound: RangeBound | None) -> QuantInfo | None:
        """(Re)build the plan when the config, shape, default dtype or ranges changed."""
        if not self.is_enabled():
            self.info = None
            return None
        config = self.config.decompose()
        assert default_dtype is not None, "default_dtype must be set."
        if self.info is None or self.info.is_outdated(config, tensor_shape, default_dtype, quant_range, range_bound):
            self.info = QuantInfo.construct(config, tensor_shape, default_dtype, quant_range=quant_range,
                                            range_bound=range_bound)
        return self.info
