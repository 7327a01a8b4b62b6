"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of deepcompressor's Nunchaku converter
(deepcompressor/backend/nunchaku/utils.py:16-352, backend/nunchaku/convert.py:12-72, backend/utils.py:12-88).
Imported by tests only; the product never imports this module.
Pinned by ``tests/golden/nunchaku.npz`` (generated from the reference by ``oracle/make_golden.py``).

16-bit tensors travel as float32 arrays holding float16 / bfloat16 values; ``dt`` names the storage type."""

from __future__ import annotations

import numpy as np

import quant_oracle as qo

F32 = np.float32
WARP_N = 128
FP4 = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=F32)


def ceil_divide(x: int, d: int) -> int:
    return (x + d - 1) // d


def pad(t: np.ndarray, divisors, dims, fill=0) -> np.ndarray:
    """backend/utils.py:29-58."""
    shape = list(t.shape)
    for d, div in zip(dims, divisors):
        shape[d] = ceil_divide(shape[d], div) * div
    out = np.full(shape, fill, dtype=t.dtype)
    out[tuple(slice(0, e) for e in t.shape)] = t
    return out


class Geometry:
    """MmaWeightPackerBase (backend/utils.py:91-157) for ``bits`` and warp_n = 128."""

    def __init__(self, bits: int):
        self.bits = bits
        self.comp_n, self.comp_k = 16, 256 // bits
        self.insn_k = self.comp_k
        self.num_lanes, self.num_k_lanes, self.num_n_lanes = 32, 4, 8
        self.reg_k, self.reg_n = 32 // bits, 1
        self.k_pack_size = self.comp_k // (self.num_k_lanes * self.reg_k)
        self.n_pack_size = self.comp_n // (self.num_n_lanes * self.reg_n)
        self.mem_n, self.mem_k = WARP_N, self.comp_k
        self.num_k_packs = self.mem_k // (self.k_pack_size * self.num_k_lanes * self.reg_k)
        self.num_n_packs = self.mem_n // (self.n_pack_size * self.num_n_lanes * self.reg_n)
        self.num_k_unrolls = 2


def pack_weight(codes: np.ndarray, bits: int) -> np.ndarray:
    """NunchakuWeightPacker.pack_weight (utils.py:20-62): fragments of 128 x (256 / bits), 32 bits per lane register."""
    g = Geometry(bits)
    n, k = codes.shape
    assert codes.dtype == np.int32 and n % g.mem_n == 0 and k % (g.mem_k * g.num_k_unrolls) == 0
    w = codes.reshape(n // g.mem_n, g.num_n_packs, g.n_pack_size, g.num_n_lanes, g.reg_n,
                      k // g.mem_k, g.num_k_packs, g.k_pack_size, g.num_k_lanes, g.reg_k)
    w = np.ascontiguousarray(w.transpose(0, 5, 6, 1, 3, 8, 2, 7, 4, 9)).astype(np.uint32)
    mask = np.uint32((1 << bits) - 1)
    shifts = (np.arange(g.reg_k, dtype=np.uint32) * np.uint32(bits))
    word = ((w & mask) << shifts).sum(axis=-1, dtype=np.uint64).astype(np.uint32)
    return np.ascontiguousarray(word).view(np.int8).reshape(n, -1)


def pack_scale(scale: np.ndarray, group_size: int) -> np.ndarray:
    """utils.py:67-104 (no micro scales: those go through :func:`pack_micro_scale`)."""
    n = scale.shape[0]
    s_pack, s_lanes = 4, 32  # min(max(128 // 32, 2), 8), min(32, 128 // 4)
    s = scale.reshape(n // WARP_N, 1, s_lanes // 4, s_pack // 2, 4, 2, -1)
    s = np.ascontiguousarray(s.transpose(0, 6, 1, 2, 4, 3, 5))
    return s.reshape(-1) if group_size == -1 else s.reshape(-1, n)


_E4M3 = None


def _e4m3_table() -> np.ndarray:
    global _E4M3
    if _E4M3 is None:
        t = np.zeros(127, dtype=np.float64)  # codes 0 .. 126 (127 is NaN)
        for c in range(127):
            e, m = c >> 3, c & 7
            t[c] = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
        _E4M3 = t
    return _E4M3


def to_e4m3fn(x: np.ndarray) -> np.ndarray:
    """float -> float8_e4m3fn bytes, round to nearest even (|x| <= 448)."""
    t = _e4m3_table()
    a = np.abs(x.astype(np.float64))
    hi = np.clip(np.searchsorted(t, a, side="left"), 0, 126)
    lo = np.clip(hi - 1, 0, 126)
    dlo, dhi = a - t[lo], t[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return (code | (np.signbit(x).astype(np.uint8) << 7)).astype(np.uint8)


def pack_micro_scale(scale: np.ndarray, group_size: int = 16) -> np.ndarray:
    """utils.py:106-138: float8 e4m3 bytes, 4 groups of 16 per 64-wide instruction."""
    assert group_size == 16 and scale.max() <= 448 and scale.min() >= -448
    n = scale.shape[0]
    s = to_e4m3fn(scale).reshape(n // WARP_N, 1, 4, 4, 8, -1, 4)
    return np.ascontiguousarray(s.transpose(0, 5, 1, 4, 3, 2, 6)).reshape(-1, n)


def pack_lowrank(w: np.ndarray, down: bool) -> np.ndarray:
    """utils.py:140-185: 16 x 16 fragments of a 16-bit operand."""
    g = Geometry(4)
    reg_n, reg_k = 1, 2
    pack_n, pack_k = g.n_pack_size * g.num_n_lanes * reg_n, g.k_pack_size * g.num_k_lanes * reg_k
    w = pad(w, (pack_n, pack_k), (0, 1))
    if down:
        r, c = w.shape
        r_packs, c_packs = r // pack_n, c // pack_k
        w = w.reshape(r_packs, pack_n, c_packs, pack_k).transpose(2, 0, 1, 3)
    else:
        c, r = w.shape
        c_packs, r_packs = c // pack_n, r // pack_k
        w = w.reshape(c_packs, pack_n, r_packs, pack_k).transpose(0, 2, 1, 3)
    w = w.reshape(c_packs, r_packs, g.n_pack_size, g.num_n_lanes, reg_n, g.k_pack_size, g.num_k_lanes, reg_k)
    return np.ascontiguousarray(w.transpose(0, 1, 3, 6, 2, 5, 4, 7)).reshape(c, r)


def unpack_lowrank(w: np.ndarray, down: bool) -> np.ndarray:
    """utils.py:187-224."""
    g = Geometry(4)
    c, r = w.shape
    reg_n, reg_k = 1, 2
    pack_n, pack_k = g.n_pack_size * g.num_n_lanes * reg_n, g.k_pack_size * g.num_k_lanes * reg_k
    if down:
        r_packs, c_packs = r // pack_n, c // pack_k
    else:
        c_packs, r_packs = c // pack_n, r // pack_k
    w = w.reshape(c_packs, r_packs, g.num_n_lanes, g.num_k_lanes, g.n_pack_size, g.k_pack_size, reg_n, reg_k)
    w = np.ascontiguousarray(w.transpose(0, 1, 4, 2, 6, 5, 3, 7)).reshape(c_packs, r_packs, pack_n, pack_k)
    if down:
        return np.ascontiguousarray(w.transpose(1, 2, 0, 3)).reshape(r, c)
    return np.ascontiguousarray(w.transpose(0, 2, 1, 3)).reshape(c, r)


def pad_scale(scale: np.ndarray, group_size: int, bits: int = 4) -> np.ndarray:
    """utils.py:234-246."""
    g = Geometry(bits)
    if group_size > 0 and scale.size > scale.shape[0]:
        s = scale.reshape(scale.shape[0], 1, -1, 1)
        k_div = g.insn_k // group_size if g.insn_k == group_size * 4 else g.num_k_unrolls
        return pad(s, (WARP_N, k_div), (0, 2), fill=1)
    return pad(scale, (WARP_N,), (0,), fill=1)


def fp_quantize(x: np.ndarray) -> np.ndarray:
    """backend/utils.py:81-88: index of the nearest float4 value, the first on ties."""
    return np.abs(x[..., None] - FP4).argmin(axis=-1)


def convert_w4(weight, scale, dt, bias=None, smooth=None, lora=None, float_point=False, subscale=None):
    """convert_to_nunchaku_w4x4y16_linear_weight (utils.py:258-330)."""
    oc, ic = weight.shape
    per_tensor = scale.size == 1
    if per_tensor:
        scale = np.broadcast_to(scale.reshape(-1), (oc,)).reshape(oc, 1, 1, 1)
    ng = scale.shape[2]
    gs = ic // ng
    nsg, sgs = (subscale.shape[2], ic // subscale.shape[2]) if subscale is not None else (ng, gs)
    w = (weight.astype(F32).reshape(oc, 1, ng, gs) / scale.astype(F32)).astype(F32)
    if subscale is not None:
        w = (w.reshape(oc, 1, nsg, sgs) / subscale.astype(F32)).astype(F32)
    w = w.reshape(oc, ic)
    if float_point:
        codes = fp_quantize(w)
    else:
        codes = np.rint(w)
        assert codes.min() >= -8 and codes.max() <= 7
    g = Geometry(4)
    bias = np.zeros((oc, 1), F32) if bias is None else bias.reshape(-1, 1)
    smooth = np.ones((ic, 1), F32) if smooth is None else smooth.reshape(-1, 1)
    codes = pad(codes.astype(np.int32), (g.mem_n, g.mem_k * g.num_k_unrolls), (0, 1))
    scale = pad_scale(scale.astype(F32), gs)
    if subscale is not None:
        subscale = pad_scale(subscale.astype(F32), sgs)
    bias, smooth = pad_scale(bias.astype(F32), -1), pad_scale(smooth.astype(F32), -1)
    packed = pack_weight(codes, 4)

    def pack_any(s, group):
        if group > 0 and g.insn_k == group * 4:
            return pack_micro_scale(s, group)
        return pack_scale(s, group)

    scale = pack_any(scale, gs if gs < ic else -1)
    if subscale is not None:
        subscale = pack_any(subscale, sgs if sgs < ic else -1)
    bias, smooth = pack_scale(bias, -1), pack_scale(smooth, -1)
    if lora is not None:
        lora = (pack_lowrank(pad(lora[0], (WARP_N,), (1,)), down=True),
                pack_lowrank(pad(lora[1], (WARP_N,), (0,)), down=False))
    if per_tensor:
        scale = scale.reshape(-1)[:1]
    return packed, scale, bias, smooth, lora, subscale


def linear_state_dict(weight, scale, dt, bias=None, smooth=None, lora=None, shift=None, smooth_fused=False,
                      float_point=False, subscale=None):
    """convert_to_nunchaku_w4x4y16_linear_state_dict (convert.py:12-72)."""
    if scale.size > 1:
        scale = scale.reshape(scale.shape[0], 1, scale.shape[2], 1)
        scale_key = "wcscales" if scale.shape[2] == 1 else "wscales"
    else:
        scale_key = "wtscale"
    subscale_key = ""
    if subscale is not None:
        subscale = subscale.reshape(subscale.shape[0], 1, subscale.shape[2], 1)
        subscale_key = "wcscales" if subscale.shape[2] == 1 else "wscales"
    if lora is not None and (smooth is not None or shift is not None):
        down, up = lora
        down = down.astype(np.float64)
        if smooth is not None and not smooth_fused:
            down = down / smooth.astype(np.float64)[None, :]
        if shift is not None:
            b = np.zeros(up.shape[0]) if bias is None else bias.astype(np.float64)
            sh = (np.broadcast_to(shift.reshape(1, 1), (down.shape[1], 1)) if shift.size == 1
                  else shift.reshape(-1, 1)).astype(np.float64)
            bias = qo.round_to(dt, (b + (up.astype(np.float64) @ down @ sh).reshape(-1)).astype(F32))
        lora = (qo.round_to(dt, down.astype(F32)), up)
    packed, scale, bias, smooth, lora, subscale = convert_w4(weight, scale, dt, bias, smooth, lora, float_point, subscale)
    out = {"qweight": packed, scale_key: scale}
    if subscale is not None:
        out[subscale_key] = subscale
    out["bias"], out["smooth_orig"] = bias, smooth
    out["smooth"] = np.ones_like(smooth) if smooth_fused else smooth.copy()
    if lora is not None:
        out["lora_down"], out["lora_up"] = lora
    return out


def convert_w8(weight, scale, bias=None):
    """convert_to_nunchaku_w8x8y16_linear_weight (utils.py:333-352)."""
    oc, ic = weight.shape
    scale = np.broadcast_to(scale.reshape(-1), (oc,)).reshape(oc, 1) if scale.size == 1 else scale.reshape(oc, 1)
    codes = np.rint((weight.astype(F32) / scale.astype(F32)).astype(F32)).astype(np.int32)
    assert codes.min() >= -128 and codes.max() <= 127
    g = Geometry(8)
    bias = np.zeros((oc, 1), F32) if bias is None else bias.reshape(-1, 1)
    packed = pack_weight(pad(codes, (g.mem_n, g.mem_k * g.num_k_unrolls), (0, 1)), 8)
    return packed, pack_scale(pad_scale(scale.astype(F32), -1, 8), -1), pack_scale(pad_scale(bias.astype(F32), -1, 8), -1)
