"""Per-tensor quantization plan: group geometry, scale levels, ranges and native step descriptors.

Mirrors the attribute names of the reference's QuantScaleInfo / QuantStepInfo / QuantInfo
(deepcompressor/quantizer/impl/scale.py:67-140, impl/info.py:18-176) so code that inspects
``quantizer.update(...).steps[-1].tensor_view_shape`` etc. keeps working; the arithmetic itself is in
the HIP kernels that ``QuantScaleInfo.quantize`` and ``QuantizerImpl`` call.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

from ... import engine
from ..._lib import Step
from ...data.dtype import QuantDataType
from ...data.range import DynamicRange, ProtectiveQuantRange, QuantRange, RangeBound
from ...data.scale import QuantScale
from ...data.utils import ScaleUtils, ShapeUtils
from ...data.zero import ZeroPointDomain
from ..config.base import DecomposedQuantizerConfig, QuantizerConfig

__all__ = ["QuantScaleInfo", "QuantStepInfo", "QuantInfo"]


@dataclass
class QuantScaleInfo:
    tensor_view_shape: torch.Size
    tensor_quant_dtype: torch.dtype | QuantDataType
    tensor_zero_domain: ZeroPointDomain | None
    tensor_quant_range: QuantRange
    tensor_range_bound: RangeBound | None
    default_quant_dtype: torch.dtype | QuantDataType
    scale_view_shapes: list[torch.Size]
    scale_quant_dtypes: list[torch.dtype | QuantDataType]
    user_quant_range: QuantRange | None = field(init=False, default=None)

    has_zero_point = property(lambda self: self.tensor_zero_domain is not None)

    def __post_init__(self):
        if isinstance(self.tensor_quant_dtype, torch.dtype):
            raise NotImplementedError("torch.dtype is not supported yet.")
        self.user_quant_range = self.tensor_quant_range
        self.tensor_quant_range = QuantRange.construct(self.tensor_quant_dtype, has_zero_point=self.has_zero_point,
                                                       quant_range=self.tensor_quant_range)
        self.scale_quant_dtypes = ScaleUtils.infer_scale_dtypes(self.scale_quant_dtypes, self.default_quant_dtype)
        self.exponent_scale_level = ScaleUtils.infer_exponent_scale_level(self.scale_quant_dtypes)
        qr = self.tensor_quant_range
        if self.has_zero_point:
            if self.tensor_zero_domain == ZeroPointDomain.PreScale:
                self.zero_quant_dtype = self.tensor_quant_dtype
            elif self.tensor_zero_domain == ZeroPointDomain.PostScale:
                self.zero_quant_dtype = self.scale_quant_dtypes[-1]
                if isinstance(self.zero_quant_dtype, QuantDataType) and self.zero_quant_dtype.is_exponent:
                    self.zero_quant_dtype = self.default_quant_dtype
            else:
                raise ValueError(f"Unsupported zero point domain: {self.tensor_zero_domain}")
            self.linear_tensor_quant_span = qr.max - qr.min
            self.exponent_tensor_quant_span = 2 ** int(math.log2(qr.max) + int(self.tensor_quant_dtype.signed))
        else:
            self.zero_quant_dtype = None
            self.linear_tensor_quant_span = qr.max
            self.exponent_tensor_quant_span = 2 ** int(math.log2(qr.max))
        n = len(self.scale_quant_dtypes)
        split = self.exponent_scale_level if 0 <= self.exponent_scale_level < n else n
        self.linear_scale_quant_dtypes = self.scale_quant_dtypes[:split]
        self.linear_scale_view_shapes = self.scale_view_shapes[:split]
        self.exponent_scale_quant_dtypes = self.scale_quant_dtypes[split:]
        self.exponent_scale_view_shapes = self.scale_view_shapes[split:]
        if self.exponent_scale_quant_dtypes:
            self.exponent_scale_quant_spans = ScaleUtils.infer_scale_quant_spans(self.exponent_scale_quant_dtypes)
            self.linear_scale_quant_spans = (
                ScaleUtils.infer_scale_quant_spans(self.linear_scale_quant_dtypes,
                                                   base=self.exponent_scale_quant_spans[-1])
                if self.linear_scale_quant_dtypes else [])
        else:
            self.exponent_scale_quant_spans = []
            self.linear_scale_quant_spans = ScaleUtils.infer_scale_quant_spans(self.linear_scale_quant_dtypes)
        self._native: dict[torch.device, Step] = {}

    # ---- native descriptor -----------------------------------------------------------------
    @property
    def num_groups(self) -> int:
        return math.prod(self.tensor_view_shape[0::2])

    @property
    def group_size(self) -> int:
        return math.prod(self.tensor_view_shape[1::2])

    def is_single_level(self) -> bool:
        return len(self.scale_quant_dtypes) == 1 and not self.exponent_scale_quant_dtypes

    def is_two_level_per_tensor(self) -> bool:
        """Two linear scale levels and no power-of-two ones: one scale for the whole tensor (a float type), then one
        per group relative to it -- the NVFP4 layout.  No zero point."""
        if self.has_zero_point or self.exponent_scale_quant_dtypes or len(self.linear_scale_quant_dtypes) != 2:
            return False
        first, second = self.linear_scale_view_shapes
        return (math.prod(first[0::2]) == 1 and math.prod(second[1::2]) == 1
                and isinstance(self.linear_scale_quant_dtypes[0], torch.dtype))

    def native(self, device: torch.device) -> Step:
        """The dcq_step of this step on ``device`` (codebook tables live on the device)."""
        if device not in self._native:
            self._native[device] = engine.make_step(
                quant_dtype=self.tensor_quant_dtype, zero_domain=self.tensor_zero_domain,
                tensor_quant_range=self.tensor_quant_range, quant_range=self.user_quant_range,
                scale_dtype=self.scale_quant_dtypes[-1], zero_dtype=self.zero_quant_dtype,
                range_bound=self.tensor_range_bound, q_span=self.linear_tensor_quant_span, device=device)
        return self._native[device]

    def _native_exponent(self, device: torch.device) -> Step:
        """The same step with the exponent levels' tensor span: range / span is then the un-quantized scale the
        power-of-two levels start from (quantizer/impl/scale.py:204-206)."""
        key = ("exp", device)
        if key not in self._native:
            self._native[key] = engine.make_step(
                quant_dtype=self.tensor_quant_dtype, zero_domain=self.tensor_zero_domain,
                tensor_quant_range=self.tensor_quant_range, quant_range=self.user_quant_range,
                scale_dtype=self.scale_quant_dtypes[-1], zero_dtype=self.zero_quant_dtype,
                range_bound=self.tensor_range_bound, q_span=self.exponent_tensor_quant_span, device=device)
        return self._native[key]

    # ---- scale tree on canonical arrays ----------------------------------------------------
    def _quantize_levels(self, lin: torch.Tensor, views=None, dtypes=None, spans=None) -> QuantScale:
        """quantize_scale: ``lin`` is the un-quantized scale, one value per group in canonical order; the levels
        default to the linear ones."""
        views = self.linear_scale_view_shapes if views is None else views
        dtypes = self.linear_scale_quant_dtypes if dtypes is None else dtypes
        spans = self.linear_scale_quant_spans if spans is None else spans
        device, scale = lin.device, QuantScale()
        finest = self.scale_view_shapes[-1]
        keep = [s for s in finest[0::2]]  # (#g0, #g1, ...)
        s = lin
        n_levels = len(dtypes)
        for level, (view, dtype, span) in enumerate(zip(views, dtypes, spans, strict=True)):
            rounding = engine.make_rounding(dtype, has_zero_point=False, quant_range=None, device=device)
            last = level == n_levels - 1
            children = math.prod(view[1::2])
            if last and children == 1:
                assert span == 1, "The last quant span must be 1."
                leaf = engine.simple_round(rounding, s).view([v for pair in zip(keep, [1] * len(keep)) for v in pair])
            else:
                parents = math.prod(view[0::2])
                n = len(view)
                perm = list(range(0, n, 2)) + list(range(1, n, 2))
                s_pq = s.view(view).permute(perm).reshape(parents, children).contiguous()
                leaf = engine.scale_level(rounding, s_pq, parents, children, reduce=True, span=span)
                leaf = leaf.view([v for pair in zip(view[0::2], [1] * (n // 2)) for v in pair])
                inverse = [perm.index(i) for i in range(n)]
                s = s_pq.view(*view[0::2], *view[1::2]).permute(inverse).reshape(-1).contiguous()
            scale.append(leaf)
        # zero scales become 1 (only in the compound; a lone leaf is the compound)
        fixed = engine.scale_join(scale.data.contiguous().view(-1), None, fix_zero=True).view(scale.data.shape)
        if scale.num_leaves == 1:
            scale._leaves[0] = fixed
        scale.data = fixed
        return scale

    def quantize(self, *, scale: torch.Tensor | None = None, zero: torch.Tensor | None = None,
                 tensor: torch.Tensor | None = None, dynamic_range: DynamicRange | None = None,
                 outer: torch.Tensor | None = None, outer_range: QuantRange | None = None
                 ) -> tuple[QuantScale, torch.Tensor]:
        """Scale tree and zero point of ``tensor`` (or from a given ``scale``), as QuantScale + zero tensor.

        ``outer`` (one value per group of this step, canonical order) and ``outer_range`` describe a previous
        progressive step: the measured data is clamp(tensor / outer, outer_range).
        """
        keep_view = [v if i % 2 == 0 else 1 for i, v in enumerate(self.tensor_view_shape)]
        has_exp, has_lin = bool(self.exponent_scale_quant_dtypes), bool(self.linear_scale_quant_dtypes)
        exp_raw = None  # range / exponent span (range-based) or the given scale: what the exponent levels quantize
        vmin = None
        if scale is None:
            assert isinstance(tensor, torch.Tensor), "View tensor must be a tensor."
            step = self.native(tensor.device)
            dr = dynamic_range or DynamicRange()
            static = dr.ratio is None and dr.max is not None
            fmt = lambda v: None if v is None else DynamicRange._format_m_(  # noqa: E731
                v, shape=torch.Size(keep_view), dtype=torch.float32, device=tensor.device).contiguous().view(-1)
            vmax = vavg = None
            if not static:
                x2d, _ = engine.canonical_groups(tensor, self.tensor_view_shape)
                lo, hi = (outer_range.min, outer_range.max) if outer is not None else (0.0, 0.0)
                vmax, vmin = engine.group_range(x2d, with_min=self.has_zero_point, outer=outer, outer_lo=lo,
                                                outer_hi=hi)
                if self.has_zero_point and self.tensor_quant_dtype.is_float_point:
                    # range symmetric about the group mean (data/range.py:323-341); the mean is a plain reduction
                    if outer is not None:
                        raise NotImplementedError("floating-point dtype with a zero point after a progressive step")
                    vavg = x2d.to(torch.float32).mean(dim=1).contiguous()
            ratio = dr.ratio
            if isinstance(ratio, torch.Tensor):
                if ratio.numel() == 1:
                    ratio = float(ratio)
                else:
                    # one ratio per group, per group of channels ... (what a range search on a non-static tensor
                    # stores, calib/range.py:164): DynamicRange.measure's `vmax * self.ratio` with broadcasting
                    # (data/range.py:331-344), on the per-group arrays
                    assert not static, "a ratio tensor scales a measured range"
                    per_group = ratio.to(device=tensor.device, dtype=torch.float32).expand(keep_view).reshape(-1)
                    if vavg is not None:
                        vmag = torch.maximum(vmax - vavg, vavg - vmin) * per_group
                        vmax, vmin, vavg = vavg + vmag, vavg - vmag, None
                    else:
                        vmax = vmax * per_group
                        vmin = vmin * per_group if vmin is not None else None
                    ratio = None
            if has_exp:
                exp_raw, _ = engine.range_to_scale(self._native_exponent(tensor.device), vmax, vmin, ratio=ratio,
                                                   static_max=fmt(dr.max),
                                                   static_min=fmt(dr.min) if self.has_zero_point else None, vavg=vavg)
            lin, vmin = engine.range_to_scale(step, vmax, vmin, ratio=ratio, static_max=fmt(dr.max),
                                              static_min=fmt(dr.min) if self.has_zero_point else None, vavg=vavg)
        else:
            lin = scale.to(torch.float32).reshape(-1).contiguous()
            assert lin.numel() == self.num_groups, "scale must hold one value per quantization group"
            if has_exp:  # scale.py:189-190, 207-208
                exp_raw = lin
                if has_lin:
                    lin = lin.mul(self.exponent_tensor_quant_span).div(self.linear_tensor_quant_span)
        s = self._quantize_levels(lin) if has_lin else QuantScale()
        if has_exp:  # scale.py:204-221: the power-of-two levels quantize what the linear levels left over
            if s.data is not None:
                s.data = s.data.expand(self.linear_scale_view_shapes[-1]).reshape(self.scale_view_shapes[-1])
                exp_raw = exp_raw / s.data.reshape(-1)
            exp_s = self._quantize_levels(exp_raw.contiguous(), self.exponent_scale_view_shapes,
                                          self.exponent_scale_quant_dtypes, self.exponent_scale_quant_spans)
            s = exp_s if s.data is None else s.extend(exp_s)
        if self.has_zero_point:
            device = s.data.device
            step = self.native(device)
            flat = s.data.contiguous().view(-1)
            if scale is None:
                z = engine.zero_point(step, vmin, flat).view(keep_view)
            else:
                assert isinstance(zero, torch.Tensor), "Zero point must be a tensor."
                z = engine.simple_round_signed(step.zero, zero.to(torch.float32).contiguous()).view(keep_view)
        else:
            z = torch.tensor(0, dtype=s.data.dtype, device=s.data.device)
        return s, z


@dataclass
class QuantStepInfo:
    quant_dtype: QuantDataType
    zero_domain: ZeroPointDomain | None
    group_shapes: tuple[tuple[int, ...], ...]
    scale_dtypes: tuple[torch.dtype | QuantDataType | None, ...]
    quant_range: QuantRange | None
    range_bound: RangeBound | None
    default_dtype: torch.dtype
    tensor_shape: torch.Size
    tensor_group_shapes: list[torch.Size]
    tensor_view_shape: torch.Size
    scale: QuantScaleInfo = field(init=False)

    def __post_init__(self):
        self.scale = QuantScaleInfo(
            tensor_view_shape=self.tensor_view_shape, tensor_quant_dtype=self.quant_dtype,
            tensor_zero_domain=self.zero_domain, tensor_quant_range=self.quant_range,
            tensor_range_bound=self.range_bound,
            scale_view_shapes=ShapeUtils.infer_scale_view_shapes(self.tensor_group_shapes, shape=self.tensor_shape),
            scale_quant_dtypes=self.scale_dtypes, default_quant_dtype=self.default_dtype)

    tensor_zero_domain = property(lambda self: self.scale.tensor_zero_domain)
    tensor_quant_range = property(lambda self: self.scale.tensor_quant_range)
    tensor_range_bound = property(lambda self: self.scale.tensor_range_bound)

    def to_config(self) -> QuantizerConfig:
        return QuantizerConfig(dtype=self.quant_dtype, zero_point=self.zero_domain,
                               group_shapes=self.tensor_group_shapes, scale_dtypes=self.scale.scale_quant_dtypes)

    @staticmethod
    def construct(config: QuantizerConfig, tensor_shape: torch.Size, default_dtype: torch.dtype,
                  quant_range: QuantRange | None = None, range_bound: RangeBound | None = None) -> "QuantStepInfo":
        group_shapes = ShapeUtils.infer_group_shapes(config.group_shapes, shape=tensor_shape)
        return QuantStepInfo(
            quant_dtype=config.dtype, zero_domain=config.zero_point, group_shapes=config.group_shapes,
            scale_dtypes=config.scale_dtypes, quant_range=quant_range, range_bound=range_bound,
            default_dtype=default_dtype, tensor_shape=tensor_shape, tensor_group_shapes=group_shapes,
            tensor_view_shape=ShapeUtils.infer_view_shape(tensor_shape, group_shape=group_shapes[-1]))


@dataclass
class QuantInfo:
    steps: tuple[QuantStepInfo, ...]
    needs_dequant_saturation: bool = False

    num_steps = property(lambda self: len(self.steps))

    def get_child(self, idx: int) -> QuantStepInfo:
        return self.steps[idx]

    def is_outdated(self, config: DecomposedQuantizerConfig, tensor_shape: torch.Size, default_dtype: torch.dtype,
                    quant_range: QuantRange | None = None, range_bound: RangeBound | None = None) -> bool:
        if self.num_steps != config.num_steps:
            return True
        for info, cfg in zip(self.steps, config.steps, strict=True):
            if (info.quant_dtype != cfg.quant_dtype or info.group_shapes != cfg.group_shapes
                    or info.scale_dtypes != cfg.scale_dtypes or info.zero_domain != cfg.zero_domain):
                return True
        if self.num_steps > 0:
            first = self.steps[0]
            if (first.tensor_shape != tensor_shape or first.default_dtype != default_dtype
                    or first.range_bound != range_bound or self.steps[-1].quant_range != quant_range):
                return True
            if self.num_steps > 1 and self.needs_dequant_saturation != config.needs_dequant_saturation:
                return True
        return False

    @staticmethod
    def construct(config: DecomposedQuantizerConfig, tensor_shape: torch.Size, default_dtype: torch.dtype,
                  quant_range: QuantRange | None = None, range_bound: RangeBound | None = None) -> "QuantInfo":
        steps: list[QuantStepInfo] = []
        n = config.num_steps
        step_default, step_bound = default_dtype, range_bound
        for i, cfg in enumerate(config.steps):
            assert cfg.quant_dtype is not None, f"quant_dtype is required for step {i}"
            if i == n - 1:
                step_range = quant_range
            elif i < n - 2 or config.needs_dequant_saturation:
                step_range = None
            else:  # only the step right before the last one can be protected instead of saturated
                step_range = ProtectiveQuantRange.construct(
                    outer_dtype=cfg.quant_dtype, inner_dtype=config.steps[-1].quant_dtype,
                    zero_domain=config.steps[-1].zero_domain, inner_quant_range=quant_range)
            steps.append(QuantStepInfo.construct(cfg, tensor_shape=tensor_shape, default_dtype=step_default,
                                                 quant_range=step_range, range_bound=step_bound))
            step_default = cfg.quant_dtype
            step_bound = steps[-1].scale.tensor_quant_range
        return QuantInfo(steps=tuple(steps), needs_dequant_saturation=config.needs_dequant_saturation)
