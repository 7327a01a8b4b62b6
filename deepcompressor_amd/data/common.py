"""Tensor roles (reference: deepcompressor/data/common.py:9-14)."""

import enum

__all__ = ["TensorType"]


class TensorType(enum.Enum):
    Weights = enum.auto()
    Inputs = enum.auto()
    Outputs = enum.auto()
