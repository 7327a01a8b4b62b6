"""Zero-point domain (reference: deepcompressor/data/zero.py:9-13)."""

import enum

__all__ = ["ZeroPointDomain"]


class ZeroPointDomain(enum.Enum):
    PreScale = enum.auto()  # q = round(x / s + z),  x' = (q - z) * s
    PostScale = enum.auto()  # q = round((x + z) / s),  x' = q * s - z
