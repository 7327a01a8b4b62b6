"""dtype helpers (reference: deepcompressor/data/utils/dtype.py:13-110)."""

from __future__ import annotations

import torch

from ..dtype import QuantDataType

__all__ = ["infer_dtype_bits", "infer_dtype_name", "eval_dtype"]

_BITS = {torch.float64: 64, torch.float32: 32, torch.float16: 16, torch.bfloat16: 16, torch.int32: 32,
         torch.int16: 16, torch.int8: 8, torch.uint8: 8}
_NAMES = {torch.float16: "fp16", torch.float32: "fp32", torch.float64: "fp64", torch.bfloat16: "bf16"}
_ALIASES = {"fp16": torch.float16, "half": torch.float16, "float16": torch.float16, "fp32": torch.float32,
            "float": torch.float32, "float32": torch.float32, "fp64": torch.float64, "double": torch.float64,
            "float64": torch.float64, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "int8": torch.int8,
            "uint8": torch.uint8, "int16": torch.int16, "int32": torch.int32}


def infer_dtype_bits(dtype: torch.dtype | QuantDataType) -> int:
    if isinstance(dtype, QuantDataType):
        return dtype.total_bits
    if dtype not in _BITS:
        raise ValueError(f"Unknown dtype {dtype}")
    return _BITS[dtype]


def infer_dtype_name(dtype: torch.dtype | QuantDataType) -> str:
    if isinstance(dtype, QuantDataType):
        return str(dtype)
    if isinstance(dtype, torch.dtype):
        return _NAMES.get(dtype, str(dtype).split(".")[-1])
    raise ValueError(f"Unknown dtype {dtype}")


def eval_dtype(s, with_quant_dtype: bool = True, with_none: bool = True):
    """Parse 'torch.float16', 'fp16', 'sint8', 'None', ... into a torch.dtype / QuantDataType / None."""
    if isinstance(s, torch.dtype):
        return s
    if isinstance(s, QuantDataType):
        if with_quant_dtype:
            return s
        raise ValueError(f"Unknown dtype {s}")
    if s is None:
        if with_none:
            return None
        raise ValueError(f"Unknown dtype {s}")
    assert isinstance(s, str), f"Unknown dtype {s}"
    text = s.lower()
    if text in ("", "none", "null", "nil"):
        if with_none:
            return None
        raise ValueError(f"Unknown dtype {s}")
    if text.startswith("torch."):
        text = text[6:]
    if text in _ALIASES:
        return _ALIASES[text]
    if with_quant_dtype:
        return QuantDataType.from_str(text)
    raise ValueError(f"Unknown dtype {s}")
