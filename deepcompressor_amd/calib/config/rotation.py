"""Rotation configuration (reference: deepcompressor/calib/config/rotation.py:16-60)."""

from __future__ import annotations

import os
from dataclasses import dataclass, field

from ...utils.config import configclass

__all__ = ["QuantRotationConfig"]


@configclass
@dataclass
class QuantRotationConfig:
    """``random``: a random Hadamard sample instead of the fixed Hadamard matrix; ``transforms``: keys of the modules
    that get an explicit Hadamard transform (``out_proj``: per-head rotation between v_proj and o_proj;
    ``down_proj``: an online Hadamard transform of the down projection's input); ``path`` + ``name``: a stored
    rotation matrix."""

    name: str = ""
    path: str = ""
    random: bool = False
    transforms: list[str] = field(default_factory=list)

    def __post_init__(self) -> None:
        self.transforms = sorted(set(self.transforms or []))
        if self.path and os.path.exists(self.path):
            assert self.name, "The name of the rotation quantization configuration must be provided."
            self.random = False
        else:
            self.path = ""
            self.name = "random" if self.random else "hadamard"

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        name = self.name
        if self.transforms:
            name += f".[{'+'.join(self.transforms)}]"
        return [f"{prefix}.{name}" if prefix else name]
