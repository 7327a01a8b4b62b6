"""Tree of per-level / per-step scales (reference: deepcompressor/data/scale.py:11-117)."""

from __future__ import annotations

import typing as tp

import torch

__all__ = ["QuantScale"]


def _compound(outer: torch.Tensor | None, inner: torch.Tensor) -> torch.Tensor:
    """inner (finer, shape (#g0, 1, #g1, 1, ...)) times outer broadcast over its children."""
    if outer is None:
        return inner
    split = []
    for d in range(0, outer.ndim, 2):
        split += [outer.shape[d], inner.shape[d] // outer.shape[d]]
    return inner.view(split).mul(outer).view(inner.shape)


class QuantScale:
    data: torch.Tensor

    def __init__(self):
        self.data, self._children, self._leaves = None, [], []

    num_children = property(lambda self: len(self._children))
    num_leaves = property(lambda self: len(self._leaves))

    def is_quantized(self) -> bool:
        return self.data is not None and bool(self._leaves or all(c.is_quantized() for c in self._children))

    def get_child(self, index: int) -> "QuantScale":
        return self._children[index]

    def append(self, scale: tp.Union[torch.Tensor, "QuantScale"]) -> "QuantScale":
        if isinstance(scale, torch.Tensor):
            assert not self._children, "a scale holds either tensors or child scales"
            self.data = _compound(self.data, scale)
            self._leaves.append(scale)
        elif isinstance(scale, QuantScale):
            assert not self._leaves, "a scale holds either tensors or child scales"
            self.data = _compound(self.data, scale.data)
            self._children.append(scale)
        else:
            raise TypeError(f"Unsupported scale type: {type(scale)}")
        return self

    def extend(self, scale: "QuantScale") -> "QuantScale":
        self.data = _compound(self.data, scale.data)
        if scale._children:
            assert not self._leaves
            self._children.extend(scale._children)
        elif scale._leaves:
            self._leaves.extend(scale._leaves)
        return self

    def join(self, scale: "QuantScale") -> "QuantScale":
        return QuantScale().append(self).append(scale)

    def remove_zero(self) -> "QuantScale":
        self.data[self.data == 0] = 1
        return self

    def state_dict(self, param_name: str, device: torch.device | str = "cpu", flatten: bool = True,
                   level_base: int = 0) -> dict[str, torch.Tensor]:
        """``{param_name}.{level}`` -> tensor, levels numbered across steps when ``flatten``."""
        if not self._children:
            return {f"{param_name}.{level_base + i}": leaf.to(device) for i, leaf in enumerate(self._leaves)}
        out: dict[str, torch.Tensor] = {}
        for i, child in enumerate(self._children):
            name = param_name if flatten else f"{param_name}.{i}"
            out.update(child.state_dict(name, device, flatten, len(out) if flatten else 0))
        return out
