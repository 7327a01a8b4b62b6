"""Small shared helpers (reference: deepcompressor/utils/common.py)."""

from __future__ import annotations

import typing as tp

import torch

__all__ = ["num2str", "tree_map"]


def num2str(num: int | float) -> str:
    """1 -> '1', 0.01 -> '0p01', -2.5 -> 'n2p5' (used in cache directory names)."""
    whole, _, frac = str(num).replace("-", "n").partition(".")
    return whole if not frac or int(frac) == 0 else f"{whole}p{frac}"


def tree_map(func: tp.Callable[[tp.Any], tp.Any], tree: tp.Any) -> tp.Any:
    """Apply ``func`` to every tensor of a nested dict / list / tuple; other leaves pass through."""
    if isinstance(tree, dict):
        return {k: tree_map(func, v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(func, v) for v in tree)
    if isinstance(tree, (torch.Tensor, torch.nn.Module)):
        return func(tree)
    return tree
