"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of deepcompressor's TinyChat (AWQ W4A16) converter
(deepcompressor/backend/tinychat/utils.py:11-107) and of Nunchaku's W4A16 converter built on it
(deepcompressor/backend/nunchaku/utils.py:341-368).  Imported by tests only; the product never imports this module.
Pinned by ``tests/golden/tinychat.npz`` (generated from the reference by ``oracle/make_golden.py``).

16-bit tensors travel as float32 arrays holding float16 / bfloat16 values; ``dt`` names the storage type."""

from __future__ import annotations

import numpy as np

import quant_oracle as qo

F32 = np.float32


def ceil_num_groups(in_features: int, group_size: int) -> int:
    """utils.py:11-42 for 4-bit weights: whole packs of 8 groups, 128 / group_size packs at a time."""
    packs = -(-(in_features // group_size) // 8)
    factor = 1 if group_size >= 128 else {64: 2, 32: 4}[group_size]
    return -(-packs // factor) * factor * 8


def pack_w4(codes: np.ndarray) -> np.ndarray:
    """utils.py:45-54."""
    oc, ic = codes.shape
    w = codes.astype(np.int32).reshape(-1, 4, 8)
    w = w[:, 0] | (w[:, 1] << 4) | (w[:, 2] << 8) | (w[:, 3] << 12)
    w = w.reshape(oc // 4, 4, ic // 64, 16).transpose(0, 2, 1, 3).reshape(oc // 4, ic)
    return w.astype(np.uint16).view(np.int16)


def convert(weight, scale, zero, dt: str, zero_pre_scaled: bool = False):
    """convert_to_tinychat_w4x16y16_linear_weight (utils.py:57-107)."""
    w = weight.astype(F32)
    oc, ic = w.shape
    scale, zero = scale.astype(F32), zero.astype(F32)
    if zero_pre_scaled:
        zero = (zero * scale).astype(F32)
    if scale.size == 1:
        scale, ng = np.broadcast_to(scale.reshape(1, 1), (oc, 1)), 1
    else:
        ng = scale.size // oc
    gs = ic // ng
    scale = scale.reshape(oc, ng, 1)
    zero = (np.broadcast_to(zero.reshape(1, 1), (oc, ng)) if zero.size == 1 else zero.reshape(oc, ng)).reshape(oc, ng, 1)
    codes = np.rint(((w.reshape(oc, ng, gs) + zero).astype(F32) / scale).astype(F32)).reshape(oc, ic)
    assert codes.min() >= 0 and codes.max() <= 15
    rows = ceil_num_groups(ic, gs)
    out_scale, out_zero = np.zeros((rows, oc), F32), np.zeros((rows, oc), F32)
    out_scale[:ng] = qo.round_to(dt, scale.reshape(oc, ng).T)
    out_zero[:ng] = -qo.round_to(dt, zero.reshape(oc, ng).T)
    return pack_w4(codes.astype(np.int32)), out_scale, out_zero


def convert_w4x16(weight, scale, bias, splits: int, dt: str = "bf16"):
    """convert_to_nunchaku_w4x16_linear_weight (nunchaku/utils.py:341-368) with the default zero of 7."""
    oc, ic = weight.shape
    ng = scale.shape[2]
    if splits > 1:
        weight = weight.reshape(splits, oc // splits, ic).transpose(1, 0, 2).reshape(oc, ic)
        scale = scale.reshape(splits, oc // splits, ng).transpose(1, 0, 2).reshape(oc, 1, ng, 1)
        delta = np.zeros(splits, F32)
        delta[1] = delta[-2] = 1
        bias = qo.round_to(dt, (bias.reshape(splits, oc // splits).T + delta).astype(F32)).reshape(oc)
    packed, s, z = convert(weight, scale, np.full_like(scale, 7), dt, zero_pre_scaled=True)
    return np.ascontiguousarray(packed).view(np.int32), s, z, bias
