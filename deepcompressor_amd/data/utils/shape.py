"""Group geometry (reference: deepcompressor/data/utils/shape.py:14-214).

A group shape entry <= 0 means "the whole dimension"; the last entry repeats for extra dimensions.
The quantizer views a tensor as (#g0, gs0, #g1, gs1, ...).
"""

from __future__ import annotations

import typing as tp

import torch

from ..dtype import QuantDataType

__all__ = ["infer_group_shape_name", "format_group_configs", "infer_group_shapes", "infer_view_shape",
           "infer_scale_view_shapes", "infer_shape"]


def infer_group_shape_name(group_shape: tp.Sequence[int]) -> str:
    g0, g1, rest = group_shape[0], group_shape[1], group_shape[2:]
    if all(g <= 0 for g in rest) or all(g == 1 for g in rest):
        whole = all(g <= 0 for g in rest)
        chan, grp = ("gchn", "g") if whole else ("vchn", "v")
        row = "" if g0 == 1 else (("ts" if g1 > 0 else "") if g0 <= 0 else f"t{g0}")
        if g1 <= 0:
            if g0 <= 0:
                return "tsnr" if whole else "tspx"
            return f"{row}{chan}"
        return f"{row}{grp}{g1}"
    return "x".join(str(g) if g >= 1 else "_" for g in group_shape)


def format_group_configs(*, group_shapes, scale_dtypes):
    """Normalise to tuples; 2-entry group shapes get a trailing -1; exponent scales must come last."""
    from .dtype import eval_dtype

    assert isinstance(group_shapes, (list, tuple)), "group_shapes must be a list or tuple"
    shapes = []
    for gs in group_shapes:
        if isinstance(gs, tp.Sequence):
            gs = tuple(int(g) for g in gs)
            assert len(gs) >= 2, "the group shape must have at least two dimensions"
            assert all(g >= -1 for g in gs), "the group shape must be larger than -1"
            shapes.append(gs if len(gs) >= 3 else (*gs, -1))
    dtypes = tuple(scale_dtypes) if isinstance(scale_dtypes, tp.Sequence) else (scale_dtypes,)
    dtypes = tuple(d if isinstance(d, (torch.dtype, QuantDataType, type(None))) else eval_dtype(d) for d in dtypes)
    assert len(shapes) > 0, "group_sizes must be a non-empty list"
    assert len(shapes) == len(dtypes), f"group_shapes {shapes} and scale_dtypes {dtypes} must have the same length"
    seen_linear = False
    for d in reversed(dtypes):
        if isinstance(d, QuantDataType) and d.is_exponent:
            if seen_linear:
                raise ValueError("The exponential scale must be after linear scale")
        else:
            seen_linear = True
    assert all(isinstance(d, QuantDataType) for d in dtypes[1:])
    return tuple(shapes), dtypes


def infer_group_shapes(group_shapes, shape: torch.Size) -> list[torch.Size]:
    assert isinstance(shape, torch.Size), f"shape must be torch.Size, got {shape} ({type(shape)})"
    assert len(shape) >= 2, f"shape must have at least 2 dimensions, got {shape}"
    out, parent = [], tuple(shape)
    for level, cfg in enumerate(group_shapes):
        cur = []
        for i, ts in enumerate(shape):
            gs = cfg[min(i, len(cfg) - 1)]
            gs = min(ts if gs <= 0 else gs, parent[i])
            assert parent[i] % gs == 0, f"level {level}: group size {gs} does not divide {parent[i]} (dim {i})"
            cur.append(gs)
        out.append(torch.Size(cur))
        parent = tuple(cur)
    return out


def infer_view_shape(tensor_shape: torch.Size, /, group_shape: tp.Sequence[int],
                     skip_first_dim: bool = False) -> torch.Size:
    view = []
    for i, ts in enumerate(tensor_shape):
        gs = group_shape[min(i, len(group_shape) - 1)]
        gs = ts if gs <= 0 else gs
        view += [ts // gs, gs]
    if skip_first_dim:
        view[0], view[1] = 1, tensor_shape[0]
    return torch.Size(view)


def infer_scale_view_shapes(group_shapes, shape: torch.Size) -> list[torch.Size]:
    if not isinstance(group_shapes[0], torch.Size):
        group_shapes = infer_group_shapes(group_shapes=group_shapes, shape=shape)
    finest = group_shapes[-1]
    return [torch.Size(v for ts, gs, fs in zip(shape, gshape, finest, strict=True) for v in (ts // gs, gs // fs))
            for gshape in group_shapes]


def infer_shape(view_shape: torch.Size) -> torch.Size:
    return torch.Size(view_shape[i] * view_shape[i + 1] for i in range(0, len(view_shape), 2))
