"""Dynamic-range calibration (reference: deepcompressor/calib/range.py:25-474, calib/search.py:63-837).

The reference proposes one ratio at a time (ask), fake-quantizes the whole tensor, measures the error and
keeps the best (tell).  Here the weight / tensor-error / per-group search is ONE kernel launch that evaluates
every candidate of every group from registers (``dcq_range_search_groups``); the result is identical.
"""

from __future__ import annotations

import typing as tp

import torch
import torch.nn as nn

from .. import engine
from ..data.cache import TensorsCache
from ..data.common import TensorType
from ..data.range import DynamicRange
from ..data.utils.shape import infer_view_shape
from ..quantizer.processor import Quantizer
from .config import (DynamicRangeCalibConfig, SearchBasedCalibGranularity, SearchBasedCalibObjective)

__all__ = ["DynamicRangeCalibrator", "calibrate_dynamic_range"]


class DynamicRangeCalibrator:
    """Finds the per-group (min, max) that minimises the quantization error of a tensor."""

    def __init__(self, tensor_type: TensorType, config: DynamicRangeCalibConfig, static: bool, quantizer: Quantizer,
                 pre_scale: torch.Tensor | None = None) -> None:
        assert quantizer is not None and quantizer.is_enabled(), "The tensor should be quantized."
        self.tensor_type, self.config, self.static, self.quantizer = tensor_type, config, static, quantizer
        self.pre_scale = pre_scale
        self.develop_dtype = quantizer.develop_dtype
        self.objective, self.granularity = config.objective, config.granularity
        self.ratios = config.get_ratios()
        self.num_iters = len(self.ratios)
        self.best_range = DynamicRange()
        self.best_error: torch.Tensor | None = None
        self.candidate_errors: torch.Tensor | None = None

    def is_clamp_based(self) -> bool:
        return self.static or not self.config.allow_scale

    def _unsupported(self, what: str):
        return NotImplementedError(
            f"range calibration with {what} is not implemented by the HIP search kernel "
            "(implemented: weights, TensorError objective, Group granularity, static / clamp-based range)")

    def calibrate(self, x_wgts: list[nn.Parameter] | None = None, *, record_errors: bool = False,
                  **unused) -> DynamicRange:
        """Run the whole grid search; extra reference keyword arguments (eval_inputs, modules...) are accepted
        and unused by the tensor-error objective."""
        if self.tensor_type != TensorType.Weights:
            raise self._unsupported(f"tensor type {self.tensor_type.name}")
        if self.objective != SearchBasedCalibObjective.TensorError:
            raise self._unsupported(f"objective {self.objective.name}")
        if self.granularity != SearchBasedCalibGranularity.Group:
            raise self._unsupported(f"granularity {self.granularity.name}")
        if not self.is_clamp_based():
            raise self._unsupported("allow_scale on a non-static tensor")
        if self.pre_scale is not None:
            raise self._unsupported("progressive (multi-step) quantizers")
        assert x_wgts is not None and len(x_wgts) == 1, "The weight should be a single tensor."
        w = x_wgts[0].detach().data
        q = self.quantizer
        default_dtype = w.dtype  # the search fake-quantizes with default_dtype = the weight's dtype
        info = q.update(w.shape, default_dtype=default_dtype)
        if info.num_steps != 1 or not info.steps[0].scale.is_single_level():
            raise self._unsupported("multi-level scales")
        step_info = info.steps[0]
        view = infer_view_shape(w.shape, q.config.largest_group_shape)
        assert tuple(view) == tuple(step_info.tensor_view_shape)
        x2d, _ = engine.canonical_groups(w, view)
        ratios = [r for it in self.ratios for r in it]
        out = engine.range_search_groups(step_info.scale.native(w.device), x2d, ratios, float(self.config.degree),
                                         want_errors=record_errors)
        keep = [v if i % 2 == 0 else 1 for i, v in enumerate(view)]
        self.best_range = DynamicRange(min=out["min"].view(keep) if out["min"] is not None else None,
                                       max=out["max"].view(keep), ratio=out["ratio"].view(view[::2]))
        self.best_error = out["error"].view(view[::2]).to(w.dtype)
        if record_errors:
            self.candidate_errors = out["errors"].view(len(ratios), *view[::2])
        return self.get_best()

    def get_best(self) -> DynamicRange:
        if self.static:
            return DynamicRange(min=self.best_range.min, max=self.best_range.max)
        return DynamicRange(min=self.best_range.min, max=self.best_range.max, ratio=1.0)


def calibrate_dynamic_range(tensor_type: TensorType, config: DynamicRangeCalibConfig | None, static: bool,
                            quantizer: Quantizer, modules: tp.Sequence[nn.Module], activations: TensorsCache,
                            weights: tp.Sequence[nn.Parameter] | None = None, eval_inputs: TensorsCache | None = None,
                            eval_module: nn.Module | None = None, eval_kwargs: dict[str, tp.Any] | None = None,
                            orig_weights=None, orig_activations=None, orig_eval_inputs=None
                            ) -> tp.Sequence[DynamicRange] | None:
    """Per-step dynamic ranges of ``quantizer`` (None when nothing has to be stored)."""
    if config is None or not quantizer.is_enabled():
        return None
    num_steps = quantizer.config.decompose().num_steps
    if not config.needs_search and (not static or tensor_type == TensorType.Weights):
        if config.ratio != 1.0:
            return tuple([DynamicRange(ratio=config.ratio)] + [None] * (num_steps - 1))
        return None
    if weights is None:
        weights = [m.weight for m in modules if hasattr(m, "weight")]
    if tensor_type == TensorType.Weights:
        assert len(modules) == 1 and len(weights) == 1, "only one module / weight per weight-range calibration"
    if num_steps != 1:
        raise NotImplementedError("range search for progressive (multi-step) quantizers is not implemented")
    calibrator = DynamicRangeCalibrator(tensor_type=tensor_type, config=config, static=static, quantizer=quantizer)
    return (calibrator.calibrate(x_wgts=list(weights)),)
