"""Builds the weight-stage config of an ``llm_stage`` fixture entry."""

import torch

from deepcompressor_amd.app import LlmWeightQuantizerConfig
from deepcompressor_amd.calib.config import (SearchBasedCalibGranularity, SearchBasedCalibObjective,
                                             SearchBasedCalibStrategy, SkipBasedDynamicRangeCalibConfig)
from deepcompressor_amd.data.dtype import QuantDataType
from deepcompressor_amd.data.zero import ZeroPointDomain

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def stage_config(c: dict, kernel_gptq=None) -> LlmWeightQuantizerConfig:
    sd = lambda x: None if x is None else (TORCH_DT[x] if x in TORCH_DT else QuantDataType.from_str(x))  # noqa: E731
    calib = SkipBasedDynamicRangeCalibConfig(
        degree=2.4, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
        granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0, num_grids=20,
        skips=[c["skip"]] if c.get("skip") else []) if c["search"] else None
    kw = dict(dtype=QuantDataType.from_str(c["dtype"]), zero_point=ZeroPointDomain[c["zero"]],
              group_shapes=tuple(tuple(g) for g in c["groups"]), scale_dtypes=tuple(sd(x) for x in c["sdt"]),
              calib_range=calib, kernel_gptq=kernel_gptq)
    if "inter" in c:
        kw.update(intermediate_dtypes=tuple(QuantDataType.from_str(x) for x in c["inter"]),
                  intermediate_levels=tuple(c["levels"]))
    return LlmWeightQuantizerConfig(**kw)
