"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of deepcompressor's QServe converter
(deepcompressor/backend/qserve/utils.py:10-210).  Imported by tests only; the product never imports this module.
Pinned by ``tests/golden/qserve.npz`` (generated from the reference by ``oracle/make_golden.py``)."""

from __future__ import annotations

import numpy as np

F32 = np.float32


def pack_weight(codes: np.ndarray) -> np.ndarray:
    """QServePacker.pack_weight (utils.py:18-46): tiles of 32 x 32, rows 16 apart share a byte."""
    n, k = codes.shape
    assert codes.dtype == np.uint8 and codes.max() <= 15 and n % 32 == 0 and k % 32 == 0
    w = codes.reshape(n // 32, 2, 2, 8, 1, k // 32, 1, 2, 4, 4).transpose(1, 0, 5, 6, 3, 8, 7, 2, 4, 9)
    w = np.ascontiguousarray(w)
    return ((w[1] << 4) + w[0]).astype(np.uint8).view(np.int8).reshape(n, k // 2)


def pack_scale(scale, zero, subscale):
    """QServePacker.pack_scale (utils.py:48-62)."""
    scale = scale.reshape(-1)
    n = scale.shape[0]
    if subscale is None:
        return scale, zero.reshape(-1), None
    view = (n // 32, 2, 2, 8, 1, -1)
    regroup = lambda t: np.ascontiguousarray(t.reshape(view).transpose(5, 0, 1, 3, 2, 4)).reshape(-1, n)  # noqa: E731
    return scale, regroup(zero), regroup(subscale)


def convert_w4(weight, scale, zero, subscale=None, zero_pre_scaled=False):
    """convert_to_qserve_w4x8y16_linear_weight (utils.py:64-179) on float32 arrays holding float16 values.
    Returns (packed int8, scale f16-valued, zero, subscale)."""
    w = weight.astype(F32)
    oc, ic = w.shape
    scale = np.broadcast_to(scale.astype(F32).reshape(-1), (oc,)).reshape(oc, 1).copy()
    zero = zero.astype(F32)
    if subscale is not None:
        subscale = np.rint(subscale.astype(F32).reshape(oc, -1, 1))
        ng = subscale.shape[1]
        zero = np.rint(np.broadcast_to(zero.reshape(1, 1) if zero.size == 1 else zero.reshape(oc, ng), (oc, ng))
                       .reshape(oc, ng, 1)).astype(F32)
        w = np.rint((w / scale).astype(F32))
        assert w.min() >= -128 and w.max() <= 127
        w = w.reshape(oc, ng, ic // ng)
        if not zero_pre_scaled:
            w = w + zero
        w = (w / subscale).astype(F32)
        if zero_pre_scaled:
            if zero.min() < 0:
                zero = zero + 8
            assert zero.min() >= 0 and zero.max() <= 15
            w = w + zero
            zero = zero * subscale
        elif w.min() < 0:
            w = w + 8
            zero = zero + 8 * subscale
        assert (w * subscale).min() >= 0 and (w * subscale).max() <= 255
        assert subscale.min() >= 0 and subscale.max() <= 127 and zero.min() >= 0 and zero.max() <= 255
        assert w.min() >= 0 and w.max() <= 15
        zero = (-zero).astype(np.int8)
        subscale = subscale.astype(np.int8)
    else:
        zero = np.broadcast_to(zero.reshape(-1), (oc,)).reshape(oc, 1).copy()
        if not zero_pre_scaled:
            w = w + zero
        w = np.rint((w / scale).astype(F32))
        if zero_pre_scaled:
            zero = np.rint(zero)
            if zero.min() < 0:
                zero = zero + 8
            assert zero.min() >= 0 and zero.max() <= 15
            w = w + zero
            zero = (zero * scale).astype(F32)
        elif w.min() < 0:
            w = w + 8
            zero = (zero + 8 * scale).astype(F32)
        assert w.min() >= 0 and w.max() <= 15
        zero = zero.astype(np.float16).astype(F32)
    scale = scale.astype(np.float16).astype(F32)
    packed = pack_weight(w.reshape(oc, ic).astype(np.uint8))
    scale, zero, subscale = pack_scale(scale, zero, subscale)
    return packed, scale, zero, subscale


def convert_w8(weight, scale):
    """convert_to_qserve_w8x8y16_linear_weight (utils.py:182-210)."""
    oc = weight.shape[0]
    scale = np.broadcast_to(scale.astype(F32).reshape(-1), (oc,)).reshape(oc, 1)
    w = np.rint((weight.astype(F32) / scale).astype(F32))
    assert w.min() >= -128 and w.max() <= 127
    return w.astype(np.int8), scale.reshape(oc).astype(np.float16).astype(F32)
