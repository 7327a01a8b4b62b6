"""Low-rank branch settings of SVDQuant (reference: deepcompressor/quantizer/config/lowrank.py:13-47)."""

from dataclasses import dataclass

from ...utils.common import num2str
from ...utils.config import EnableConfig, configclass

__all__ = ["QuantLowRankConfig"]


@configclass
@dataclass
class QuantLowRankConfig(EnableConfig):
    rank: int = 32
    exclusive: bool = False  # one branch per weight instead of one shared by the weights that share inputs
    compensate: bool = False  # the branch fits the quantization error instead of the weight

    def is_enabled(self) -> bool:
        return self.rank != 0

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        if not self.is_enabled():
            return []
        name = f"r{num2str(self.rank)}" + (".exclusive" if self.exclusive else "") + (
            ".compensate" if self.compensate else "")
        return [f"{prefix}.{name}" if prefix else name]
