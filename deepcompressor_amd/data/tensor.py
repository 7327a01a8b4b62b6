"""Result of a quantization call (reference: deepcompressor/data/tensor.py:11-46)."""

from __future__ import annotations

import torch

from .scale import QuantScale

__all__ = ["QuantTensor"]


class QuantTensor:
    def __init__(self, dequantized: torch.Tensor | None = None, quantized: torch.Tensor | None = None,
                 scale: QuantScale | None = None, zero: torch.Tensor | float | None = None,
                 view_shape: torch.Size | None = None):
        assert dequantized is not None or quantized is not None, "need the dequantized or the quantized tensor"
        self._dequantized, self._quantized = dequantized, quantized
        self.scale, self.zero, self.view_shape = scale, zero, view_shape

    @property
    def data(self) -> torch.Tensor | None:
        """The dequantized (fake-quantized) tensor in the input's dtype."""
        return self._dequantized

    @property
    def qdata(self) -> torch.Tensor | None:
        """The integer / codebook values as float32."""
        return self._quantized
