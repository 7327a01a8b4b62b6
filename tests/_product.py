"""Helpers that build product-side objects from the plain-python case dictionaries of the golden fixtures."""

import numpy as np
import torch

from deepcompressor_amd.data.dtype import QuantDataType
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import ProgressiveQuantizerConfig, Quantizer, QuantizerConfig

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def _sdt(x):
    if x is None:
        return None
    return TORCH_DT[x] if x in TORCH_DT else QuantDataType.from_str(x)


def build_config(case: dict):
    kw = dict(dtype=QuantDataType.from_str(case["dtype"]),
              zero_point=None if case["zero"] is None else ZeroPointDomain[case["zero"]],
              group_shapes=tuple(tuple(g) for g in case["groups"]),
              scale_dtypes=tuple(_sdt(x) for x in case["sdt"]))
    if case.get("inter"):
        return ProgressiveQuantizerConfig(**kw, intermediate_dtypes=tuple(QuantDataType.from_str(x)
                                                                          for x in case["inter"]),
                                          intermediate_levels=tuple(case["levels"]))
    return QuantizerConfig(**kw)


def build_quantizer(case: dict, **kw) -> Quantizer:
    return Quantizer(build_config(case), **kw)


def to_device(x: np.ndarray, dt: str, device="cuda") -> torch.Tensor:
    """float32 fixture array holding `dt`-representable values -> device tensor of that dtype (exact)."""
    return torch.from_numpy(np.ascontiguousarray(x)).to(TORCH_DT[dt]).to(device)


def to_np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()
