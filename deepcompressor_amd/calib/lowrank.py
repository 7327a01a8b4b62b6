"""Low-rank branch calibration (reference: deepcompressor/calib/lowrank.py:20-235).

Alternates ``L = SVD_rank(W - Q)`` and ``Q = fake_quant(W - L)`` and keeps the iterate whose evaluation-module
outputs are closest to the original ones.  Fake quantization runs on the ``Quantizer`` kernels; the evaluation
module's forward is torch on the GPU; the SVD is delegated (see ``nn/patch/lowrank.py``).
"""

from __future__ import annotations

import typing as tp

import torch
import torch.nn as nn

from ..data.cache import TensorsCache
from ..nn.patch.lowrank import LowRankBranch
from ..quantizer.processor import Quantizer
from ..utils.config import KeyEnableConfig
from .config.lowrank import QuantLowRankCalibConfig
from .config.search import SearchBasedCalibObjective

__all__ = ["QuantLowRankCalibrator"]


class QuantLowRankCalibrator:
    def __init__(self, config: QuantLowRankCalibConfig, w_quantizer: Quantizer, x_quantizer: Quantizer | None,
                 develop_dtype: torch.dtype = torch.float32) -> None:
        if isinstance(config, KeyEnableConfig):
            assert config.is_enabled_for(w_quantizer.key), "The calibrator should be enabled for the quantizer."
        else:
            assert config.is_enabled(), "The calibrator should be enabled."
        self.config, self.w_quantizer, self.x_quantizer, self.develop_dtype = config, w_quantizer, x_quantizer, develop_dtype
        assert w_quantizer is not None and w_quantizer.is_enabled(), "The tensor should be quantized."
        self.needs_x_quant_for_wgts = x_quantizer is not None and x_quantizer.is_enabled()
        self.num_iters = config.num_iters
        self.objective = config.objective

    def is_done(self) -> bool:
        return self.iter >= self.num_iters or self.early_stopped

    def reset(self, x_wgts: list[nn.Parameter]) -> None:
        self.iter = 0
        self.best_branch: LowRankBranch | None = None
        self.best_error: torch.Tensor | None = None
        self.error_history: list[float] = []
        self.early_stopped = False
        self._state_dict, self._hooks = [], []
        if len(x_wgts) > 1 and not self.config.exclusive:
            self.w = torch.cat([w.data for w in x_wgts], dim=0)
        else:
            assert len(x_wgts) == 1
            self.w = x_wgts[0].data
        if self.config.compensate:
            self.qw = torch.cat([self.w_quantizer.quantize(w.data, kernel=None, develop_dtype=self.develop_dtype).data
                                 for w in x_wgts], dim=0)
        else:
            self.qw = 0
        self.hat_ws: list = [None] * len(x_wgts)
        self.ocs = [w.shape[0] for w in x_wgts]

    def get_best(self) -> LowRankBranch:
        return self.best_branch

    def ask(self) -> LowRankBranch:
        """The next branch fits what quantization currently loses; the weights are then re-quantized around it
        (lowrank.py:118-154)."""
        branch = LowRankBranch(self.w.shape[1], self.w.shape[0], rank=self.config.rank, weight=self.w - self.qw)
        lw = branch.get_effective_weight().view(self.w.shape)
        rw = self.w - lw
        start = 0
        for idx, oc in enumerate(self.ocs):
            self.hat_ws[idx] = self.w_quantizer.quantize(rw[start:start + oc], kernel=None,
                                                         develop_dtype=self.develop_dtype).data
            start += oc
        self.qw = torch.cat(self.hat_ws, dim=0) if len(self.hat_ws) > 1 else self.hat_ws[0]
        if self.objective != SearchBasedCalibObjective.OutputsError:
            start = 0
            for idx, oc in enumerate(self.ocs):
                self.hat_ws[idx] = self.hat_ws[idx] + lw[start:start + oc]
                start += oc
        self.candidate = branch
        return branch

    def tell(self, error: torch.Tensor) -> None:
        assert error.numel() == 1, "The error should only have one value."
        self.error_history.append(float(error))
        if self.best_error is None or bool(error <= self.best_error):
            self.best_error, self.best_branch = error, self.candidate
        elif self.config.early_stop:
            self.early_stopped = True
        self.iter += 1

    def _install(self, wgts: list[nn.Parameter], mods: list[nn.Module]) -> None:
        assert len(self.hat_ws) == len(wgts) == len(mods)
        shared = self.candidate
        start = 0
        for mod, wgt, hat_w in zip(mods, wgts, self.hat_ws, strict=True):
            self._state_dict.append((wgt, wgt.data))
            wgt.data = hat_w
            if len(wgts) > 1:  # one shared down-projection, this weight's rows of the up-projection
                branch = LowRankBranch(wgt.shape[1], wgt.shape[0], rank=self.config.rank)
                branch.a = shared.a
                branch.b.to(dtype=wgt.dtype, device=wgt.device)
                branch.b.weight.data.copy_(shared.b.weight[start:start + wgt.shape[0]])
                start += wgt.shape[0]
            else:
                branch = shared
            self._hooks.append(branch.as_hook().register(mod))
        if self.needs_x_quant_for_wgts:
            self._hooks.append(self.x_quantizer.as_hook().register(mods))
        self.hat_ws = [None] * len(self.hat_ws)

    def _recover(self) -> None:
        for p, w in self._state_dict:
            p.data = w
        self._state_dict.clear()
        for h in self._hooks:
            h.remove()
        self._hooks.clear()

    @torch.inference_mode()
    def calibrate(self, x_wgts: list[nn.Parameter], x_mods: list[nn.Module] | None = None,
                  eval_inputs: TensorsCache | None = None, eval_module: nn.Module | None = None,
                  eval_kwargs: dict[str, tp.Any] | None = None, x_acts: TensorsCache | None = None,
                  **kwargs) -> LowRankBranch:
        if self.objective != SearchBasedCalibObjective.OutputsError:
            raise NotImplementedError(f"low-rank calibration with {self.objective.name} is not implemented")
        assert eval_inputs is not None and eval_module is not None, "OutputsError needs eval_inputs and eval_module"
        eval_kwargs = eval_kwargs or {}
        eval_inputs = TensorsCache({key: ipt.repartition(max_batch_size=self.config.sample_batch_size,
                                                         max_size=self.config.sample_size, standardize=False, reshape=True)
                                    for key, ipt in eval_inputs.items()})

        def forward(index: int) -> torch.Tensor:
            ipt = eval_inputs.extract(index, eval_kwargs)
            y = eval_module(*ipt.args, **ipt.kwargs)
            return y[0] if not isinstance(y, torch.Tensor) else y

        num_batches = len(eval_inputs.front().data)
        self.reset(x_wgts)
        orig = [forward(i) for i in range(num_batches)]
        while not self.is_done():
            self.ask()
            self._install(x_wgts, x_mods)
            total = None
            for i in range(num_batches):
                y = (forward(i) - orig[i]).to(self.develop_dtype).pow_(self.config.degree).sum().view(-1)
                total = y if total is None else total.add_(y)
            self._recover()
            self.tell(total)
        return self.get_best()
