"""Channel reordering for group quantization (reference: deepcompressor/calib/reorder.py:28-540).

Channels of a linear's input are permuted so that channels of similar magnitude share a quantization group.  The
per-channel statistics come from ``ChannelMetric`` (one HBM pass per tensor, ``dcq_channel_metric``); candidates
are scored on the evaluation module's outputs with the weights and inputs fake-quantized by the HIP quantizer.
"""

from __future__ import annotations

import typing as tp
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import engine
from ..data.cache import TensorsCache
from ..quantizer.processor import ProcessHook, Quantizer
from ..utils import math
from .config.reorder import ChannelOrderCalibConfig
from .config.search import SearchBasedCalibGranularity, SearchBasedCalibObjective, SearchBasedCalibStrategy
from .metric import ChannelMetric

__all__ = ["ChannelOrderCalibrator", "ChannelReorderer", "get_channel_index_from_rank", "get_channel_metric",
           "update_channel_metric", "init_channel_index_from_metric"]


@dataclass
class ChannelReorderer:
    """Gathers the channels of a module's input (or output) in the calibrated order (reorder.py:28-58)."""

    index: torch.Tensor
    channels_dim: int
    input_packager: tp.Any = None
    output_packager: tp.Any = None

    def is_enabled(self) -> bool:
        return self.index is not None

    def get_input_packager(self):
        return self.input_packager

    def get_output_packager(self):
        return self.output_packager

    def process(self, tensor: torch.Tensor) -> torch.Tensor:
        self.index = self.index.to(device=tensor.device)
        return tensor.index_select(dim=self.channels_dim, index=self.index)

    def as_hook(self, func=None, *, is_output: bool = False) -> ProcessHook:
        return ProcessHook(self, func, is_output=is_output)


def get_channel_index_from_rank(rank: torch.Tensor, num_channels: int, num_groups: int,
                                index_mode: ChannelOrderCalibConfig.ChannelIndex) -> torch.Tensor:
    """reorder.py:61-88."""
    if index_mode == ChannelOrderCalibConfig.ChannelIndex.Transpose:
        return rank.view(num_channels // num_groups, num_groups).t().reshape(-1)
    if index_mode == ChannelOrderCalibConfig.ChannelIndex.Sequential:
        return rank
    raise ValueError(f"Unsupported index mode: {index_mode}")


def get_channel_metric(inputs: TensorsCache, weights: tp.Sequence[torch.Tensor],
                       metric_mode: ChannelOrderCalibConfig.ChannelMetric, num_channels: int, num_heads: int = 1,
                       device: torch.device | str | None = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """The statistic named by ``metric_mode`` per input channel (reorder.py:91-157)."""
    name = metric_mode.name
    M = ChannelOrderCalibConfig.ChannelMetric
    if name.endswith("Product"):
        name = name[:-7]
        kw = dict(inputs=inputs, weights=weights, num_channels=num_channels, num_heads=num_heads, device=device,
                  dtype=dtype)
        return get_channel_metric(metric_mode=M[f"Inputs{name}"], **kw) * get_channel_metric(
            metric_mode=M[f"Weights{name}"], **kw)
    if name.startswith("Inputs"):
        assert inputs.num_tensors == 1, f"Only one input source is allowed, got {inputs.num_tensors}"
        name, tensors = name[6:], inputs.front().get_standardized_data(reshape=False)
    else:
        assert name.startswith("Weights")
        name, tensors = name[7:], list(weights)
    group_shape = [-1] * tensors[0].ndim
    group_shape[1] = num_channels // num_heads
    snake = "".join("_" + c.lower() if c.isupper() else c for c in name).lstrip("_")
    return getattr(ChannelMetric, snake)(tensors, num_channels, group_shape, device=device, dtype=dtype).view(num_channels)


def update_channel_metric(metric: torch.Tensor | None, inputs: TensorsCache, weights: tp.Sequence[torch.Tensor],
                          metric_mode: ChannelOrderCalibConfig.ChannelMetric, num_channels: int, num_heads: int = 1,
                          device: torch.device | str | None = None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Fold another set of tensors into a running metric (reorder.py:160-208)."""
    new = get_channel_metric(inputs=inputs, weights=weights, metric_mode=metric_mode, num_channels=num_channels,
                             num_heads=num_heads, device=device, dtype=dtype)
    if metric is None:
        return new
    return torch.maximum(metric, new) if "Max" in metric_mode.name else metric.add_(new)


def init_channel_index_from_metric(metric: torch.Tensor, /, metric_mode: ChannelOrderCalibConfig.ChannelMetric,
                                   index_mode: ChannelOrderCalibConfig.ChannelIndex, group_size: int,
                                   num_heads: int = 1, num_head_repeats: int = 1) -> torch.Tensor:
    """Channel order from a metric: ascending rank, inside each head when there are heads, shared by the query heads
    of one key / value head (reorder.py:211-265).  Equal metric values keep their channel order."""
    num_channels = metric.numel()
    num_groups = num_channels // group_size
    if num_heads > 1:
        head_channels = num_channels // num_heads
        if num_head_repeats > 1:
            unique = num_heads // num_head_repeats
            m = metric.view(unique, num_head_repeats, head_channels)
            m = m.amax(dim=1, keepdim=True) if "Max" in metric_mode.name else m.sum(dim=1, keepdim=True)
            rank = m.argsort(dim=-1, stable=True).expand(unique, num_head_repeats, -1).reshape(num_heads, -1)
        else:
            rank = metric.view(num_heads, head_channels).argsort(dim=-1, stable=True)
        rank = rank + torch.arange(0, num_channels, head_channels, dtype=torch.long, device=rank.device).view(num_heads, 1)
        index = torch.empty_like(rank)
        for head in range(num_heads):
            index[head] = get_channel_index_from_rank(rank[head], num_channels=head_channels,
                                                      num_groups=max(num_groups // num_heads, 1), index_mode=index_mode)
        return index.view(-1)
    return get_channel_index_from_rank(metric.argsort(stable=True), num_channels=num_channels, num_groups=num_groups,
                                       index_mode=index_mode)


class ChannelOrderCalibrator:
    """Picks the channel order (reorder.py:268-540).  Manual: the configured metric / index mode.  GridSearch: the
    identity order plus every (metric, index mode) pair, compared on the evaluation module's outputs."""

    def __init__(self, config: ChannelOrderCalibConfig, weight_quantizer: Quantizer | None,
                 input_quantizer: Quantizer | None, num_heads: int = 1, num_head_repeats: int = 1,
                 develop_dtype: torch.dtype = torch.float32) -> None:
        assert config.objective == SearchBasedCalibObjective.OutputsError
        assert config.granularity == SearchBasedCalibGranularity.Layer
        self.config, self.develop_dtype = config, develop_dtype
        self.w_quantizer, self.x_quantizer = weight_quantizer, input_quantizer
        if config.strategy == SearchBasedCalibStrategy.Manual:
            self.index_modes, self.metric_modes = [config.channel_index], [config.channel_metric]
        else:
            self.metric_modes = list(ChannelOrderCalibConfig.ChannelMetric.__members__.values())
            self.index_modes = list(ChannelOrderCalibConfig.ChannelIndex.__members__.values())
        self.num_index_modes, self.num_metric_modes = len(self.index_modes), len(self.metric_modes)
        self.num_heads, self.num_head_repeats = num_heads, num_head_repeats
        self.metrics, self.channel_indexes = None, None
        self.candidate = self.candidate_inverse = None
        self.candidate_id = 0
        self.error_stats_history: list[list[float]] = []

    # ---- what gets quantized while scoring (calib/search.py:96-131 with this calibrator's switches)
    @property
    def needs_w_quant(self) -> bool:
        return self.w_quantizer is not None and self.w_quantizer.is_enabled()

    @property
    def needs_x_quant(self) -> bool:
        return self.x_quantizer is not None and self.x_quantizer.is_enabled()

    @property
    def needs_w_quant_for_wgts(self) -> bool:
        return self.config.allow_w_quant and self.needs_w_quant

    @property
    def needs_x_quant_for_wgts(self) -> bool:
        return self.config.allow_x_quant and self.needs_x_quant

    @property
    def population_size(self) -> int:
        size = self.num_index_modes * self.num_metric_modes
        return size + 1 if self.config.strategy != SearchBasedCalibStrategy.Manual else size

    # ---- metrics and candidate orders
    def update_channel_metrics(self, weights: list[torch.Tensor | nn.Parameter], inputs: TensorsCache) -> None:
        weights = [w.data for w in weights]
        if self.metrics is None:
            self.num_channels = weights[0].shape[1]
            self.device = weights[0].device
            self.metrics = [None] * len(self.metric_modes)
        for i, mode in enumerate(self.metric_modes):
            self.metrics[i] = update_channel_metric(metric=self.metrics[i], inputs=inputs, weights=weights,
                                                    metric_mode=mode, num_channels=self.num_channels,
                                                    num_heads=self.num_heads, device=self.device,
                                                    dtype=self.develop_dtype)

    def init_channel_indexes(self) -> None:
        def group_size_of(quantizer, needed: bool) -> int:
            size = quantizer.config.smallest_group_shape[1] if needed else -1
            return size if size > 0 else self.num_channels

        group_size = min(group_size_of(self.x_quantizer, self.needs_x_quant),
                         group_size_of(self.w_quantizer, self.needs_w_quant))
        self.channel_indexes = [None] + [
            init_channel_index_from_metric(metric, metric_mode=metric_mode, index_mode=index_mode,
                                           group_size=group_size, num_heads=self.num_heads,
                                           num_head_repeats=self.num_head_repeats)
            for metric_mode, metric in zip(self.metric_modes, self.metrics) for index_mode in self.index_modes]
        self.arange = torch.arange(self.num_channels, dtype=torch.long, device=self.device)
        self.metrics = None

    def reset(self, x_wgts, x_acts: TensorsCache) -> None:
        if self.channel_indexes is None:
            self.update_channel_metrics(x_wgts, x_acts)
            self.init_channel_indexes()
        if self.config.strategy == SearchBasedCalibStrategy.Manual and self.channel_indexes[0] is None:
            self.channel_indexes = self.channel_indexes[1:]
        assert len(self.channel_indexes) == self.population_size
        self.baseline_errors, self.best_error, self.best_candidate_id = None, None, None
        self.error_stats_history = []
        self.candidate_id = 0
        self._state_dict: list[tuple[nn.Parameter, torch.Tensor]] = []
        self._hooks: list = []

    def get_best(self) -> torch.Tensor:
        return self.channel_indexes[self.best_candidate_id]

    def ask(self) -> torch.Tensor | None:
        index = self.channel_indexes[self.candidate_id]
        inverse = None
        if index is not None:
            inverse = torch.zeros_like(index)
            inverse[index] = self.arange.to(device=index.device)
        self.candidate, self.candidate_inverse = index, inverse
        return index

    def tell(self, errors: list[torch.Tensor]) -> None:
        """Lexicographic comparison of (outputs that got worse, minus outputs that got better, total worsening,
        total improvement, total error), all relative to the first candidate (reorder.py:427-449)."""
        errors = [tuple(math.root_(e.to(torch.float64), self.config.degree) for e in error) for error in errors]
        if self.baseline_errors is None:
            self.baseline_errors = errors
        stats = [0, 0, 0, 0, 0]
        for base, err in zip(self.baseline_errors, errors):
            for be, e in zip(base, err):
                d = e.item() - be.item()
                if e > be:
                    stats[0] += 1
                if e < be:
                    stats[1] -= 1
                stats[2] += max(d, 0)
                stats[3] += min(d, 0)
                stats[4] += e.item()
        self.error_stats_history.append(stats)
        if self.best_error is None or stats < self.best_error:
            self.best_error, self.best_candidate_id = stats, self.candidate_id
        self.candidate_id += 1

    # ---- candidate in place (reorder.py:472-540)
    def _quantize_in_candidate_order(self, tensor: torch.Tensor, dim: int, fake_quantize) -> torch.Tensor:
        """``fake_quantize`` applied to ``tensor`` with its channels (dimension ``dim``) gathered into the candidate
        order -- so the quantization groups are the candidate's -- and scattered back afterwards."""
        order = self.candidate
        if order is None:
            return fake_quantize(tensor)
        gathered = tensor.index_select(dim=dim, index=order.to(tensor.device))
        return fake_quantize(gathered).index_select(dim=dim, index=self.candidate_inverse.to(tensor.device))

    def _process_x_in_xw(self, x: torch.Tensor, channels_dim=None) -> torch.Tensor:
        """Inputs as the quantized layer would see them under the candidate order (calib/reorder.py:472-482)."""
        if not self.needs_x_quant_for_wgts:
            return x
        dim = self.x_quantizer.channels_dim if channels_dim is None else channels_dim
        return self._quantize_in_candidate_order(
            x, dim, lambda t: self.x_quantizer.quantize(t, channels_dim=dim).data)

    def _process_w_in_xw(self, w: torch.Tensor) -> torch.Tensor:
        """Weights fake-quantized under the candidate order of their input channels (calib/reorder.py:484-492)."""
        if not self.needs_w_quant_for_wgts:
            return w
        return self._quantize_in_candidate_order(
            w, 1, lambda t: self.w_quantizer.quantize(t.data, kernel=None, develop_dtype=self.develop_dtype).data)

    def _process_wgts_centric_mod(self, wgts: list[nn.Parameter], mods: list[nn.Module], *, reorder_wgts,
                                  reorder_ipt_mods, reorder_opt_mods) -> None:
        """Reorder for real -- the listed parameters along their dimension, the listed modules' inputs / outputs by
        hooks -- then fake-quantize the weights and hook the input quantizer onto ``mods``; with everything already
        in candidate order, the quantizers see contiguous groups."""
        index = self.candidate
        self._state_dict.extend([(w, w.data) for w, _ in reorder_wgts])
        if index is not None:
            for w, d in reorder_wgts:
                w.data = w.data.index_select(dim=d, index=index.to(w.device))
            for m, channels_dim, packager in reorder_ipt_mods:
                self._hooks.append(ChannelReorderer(index, channels_dim, input_packager=packager).as_hook().register(m))
            for m, channels_dim, packager in reorder_opt_mods:
                self._hooks.append(ChannelReorderer(index, channels_dim, output_packager=packager)
                                   .as_hook(is_output=True).register(m))
        backup, self.candidate = index, None  # already reordered: _process_* must not reorder again
        try:
            if self.needs_w_quant_for_wgts:
                for w in wgts:
                    w.data = self._process_w_in_xw(w.data)
            if self.needs_x_quant_for_wgts:
                self._hooks.append(self.x_quantizer.as_hook(self._process_x_in_xw, is_output=False).register(mods))
        finally:
            self._candidate_backup = backup

    def _recover_mod(self) -> None:
        for p, w in self._state_dict:
            p.data = w
        self._state_dict.clear()
        for hook in self._hooks:
            hook.remove()
        self._hooks.clear()
        self.candidate, self._candidate_backup = self._candidate_backup, None

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
                  orig_eval_inputs: TensorsCache | None = None, *, reorder_wgts: list[tuple[nn.Parameter, int]],
                  reorder_ipt_mods: list[tuple[nn.Module, int, tp.Any]],
                  reorder_opt_mods: list[tuple[nn.Module, int, tp.Any]], **unused) -> torch.Tensor:
        """The best channel order (a long tensor; ``None`` when the identity order won a search)."""
        assert all(isinstance(w, nn.Parameter) for w in x_wgts), "wgts should be a list of nn.Parameter"
        assert eval_inputs is not None, "eval_inputs should not be None when objective is OutputsError"
        assert eval_module is not None, "eval_module should not be None when OutputsError"
        x_acts = x_acts or eval_inputs
        eval_kwargs = eval_kwargs or {}
        eval_inputs, orig_eval_inputs = self._parse_inputs(eval_inputs), self._parse_inputs(orig_eval_inputs)
        self.reset(x_wgts, x_acts)
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
            for p, w in orig_x_wgts:
                p.data = w.to(device=p.data.device)
        ref = orig_eval_inputs or eval_inputs
        base = [forward(ref, i) for i in range(len(ref.front().data))]
        for p, w in saved:
            p.data = w
        while self.candidate_id < self.population_size:
            self.ask()
            self._process_wgts_centric_mod(x_wgts, x_mods, reorder_wgts=reorder_wgts,
                                           reorder_ipt_mods=reorder_ipt_mods, reorder_opt_mods=reorder_opt_mods)
            try:
                total = None
                for i, y0 in enumerate(base):
                    e = engine.product_error(forward(eval_inputs, i), y0, degree).double().sum().to(torch.float32).view(1)
                    total = e if total is None else total.add_(e)
            finally:
                self._recover_mod()
            self.tell([total])
        return self.get_best()
