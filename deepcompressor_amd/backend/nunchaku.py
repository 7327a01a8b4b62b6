"""Nunchaku (SVDQuant inference engine) weight conversion
(reference: deepcompressor/backend/nunchaku/utils.py:16-338, backend/utils.py:12-58, 81-157).

W4A4 (integer or float4 codes, per-group scales, optional two-level micro scales, low-rank branch) and W8A8 linear
tensors in the back end's fragment order.  The code regrouping + bit packing is one HIP kernel
(``dcq_pack_nunchaku``); the scale / bias / low-rank permutations are small tensor reshapes, written here from the
layout they produce: every packed tensor is a sequence of 128-channel tiles whose channels are dealt to the 32
lanes of the consuming matrix instruction.
"""

from __future__ import annotations

import ctypes as C
import typing as tp

import torch

from .._lib import DcqError, check, lib, stream_of

__all__ = ["ceil_divide", "pad", "fp_quantize", "NunchakuWeightPacker", "convert_to_nunchaku_w4x4y16_linear_weight",
           "convert_to_nunchaku_w8x8y16_linear_weight", "convert_to_nunchaku_w4x4y16_linear_state_dict"]

_HALF_TYPES = (torch.float16, torch.bfloat16)
_FP4_VALUES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)  # code = index, + 8 for the negative value


def ceil_divide(x: int, divisor: int) -> int:
    return -(-x // divisor)


def pad(tensor: torch.Tensor | None, divisor: int | tp.Sequence[int], dim: int | tp.Sequence[int],
        fill_value: float | int = 0) -> torch.Tensor | None:
    """``tensor`` with the dimensions ``dim`` grown to multiples of ``divisor`` and the new elements set to
    ``fill_value`` (always a new tensor); the tensor itself when no divisor exceeds 1 (backend/utils.py:29-58)."""
    dims = [dim] if isinstance(dim, int) else list(dim)
    divisors = [divisor] * len(dims) if isinstance(divisor, int) else list(divisor)
    assert len(dims) == len(divisors), "one divisor per dimension"
    if tensor is None or max(divisors) <= 1:
        return tensor
    target = list(tensor.shape)
    for d, div in zip(dims, divisors):
        target[d] = ceil_divide(target[d], div) * div
    if target == list(tensor.shape):
        return tensor.clone()
    grown = tensor.new_full(target, fill_value)
    grown[tuple(slice(0, extent) for extent in tensor.shape)] = tensor
    return grown


def fp_quantize(x: torch.Tensor, codebook: torch.Tensor | None = None) -> torch.Tensor:
    """Index of the codebook entry nearest to each element, the first one on ties (backend/utils.py:81-88; default
    codebook: the sixteen float4 e2m1 values, negative ones at index + 8).  Evaluated in slabs of 4 M elements so the
    [elements][entries] distance table stays small."""
    if codebook is None:
        codebook = torch.tensor(_FP4_VALUES + tuple(-v for v in _FP4_VALUES), dtype=x.dtype, device=x.device)
    flat = x.reshape(-1)
    index = torch.empty(flat.shape, dtype=torch.long, device=x.device)
    for begin in range(0, flat.numel(), 1 << 22):
        piece = flat[begin:begin + (1 << 22)]
        index[begin:begin + piece.numel()] = (piece[:, None] - codebook[None, :]).abs().argmin(dim=1)
    return index.view(x.shape)


class NunchakuWeightPacker:
    """Layouts of backend/nunchaku/utils.py:16-255 (``MmaWeightPackerBase(bits, warp_n=128)``).

    One matrix instruction consumes a 16 (output) x ``256 / bits`` (input) fragment of codes on 32 lanes arranged as
    8 (output) x 4 (input), every lane holding 2 x 2 registers of ``32 / bits`` codes; a warp walks 128 output
    channels, and the input dimension is unrolled twice.  The attribute names are the reference's."""

    LANES, N_LANES, K_LANES = 32, 8, 4

    def __init__(self, bits: int, warp_n: int = 128):
        assert bits in (4, 8), f"weight bits {bits} is not supported."
        assert warp_n % 16 == 0 and warp_n >= 16
        self.bits, self.warp_n = bits, warp_n
        self.num_lanes, self.num_n_lanes, self.num_k_lanes = self.LANES, self.N_LANES, self.K_LANES
        self.comp_n, self.comp_k = 16, 256 // bits
        self.insn_n, self.insn_k = 8, self.comp_k
        self.reg_n, self.reg_k = 1, 32 // bits
        self.n_pack_size = self.comp_n // (self.N_LANES * self.reg_n)
        self.k_pack_size = self.comp_k // (self.K_LANES * self.reg_k)
        self.mem_n, self.mem_k, self.num_k_unrolls = warp_n, self.comp_k, 2
        self.num_n_packs = self.mem_n // (self.n_pack_size * self.N_LANES * self.reg_n)
        self.num_k_packs = self.mem_k // (self.k_pack_size * self.K_LANES * self.reg_k)

    # ---- codes -------------------------------------------------------------------------------
    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        assert weight.dtype == torch.int32, f"quantized weight should be torch.int32, but got {weight.dtype}."
        if not weight.is_cuda:
            raise DcqError("deepcompressor_amd runs on MI355X GPUs only (there is no CPU path)")
        n, k = weight.shape
        k_tile = self.mem_k * self.num_k_unrolls
        assert n % self.mem_n == 0, f"output channel size ({n}) should be divisible by mem_n ({self.mem_n})."
        assert k % k_tile == 0, (f"input channel size ({k}) should be divisible by mem_k ({self.mem_k}) * "
                                 f"num_k_unrolls ({self.num_k_unrolls}).")
        assert self.warp_n == 128, "the pack kernel is built for warp_n = 128"
        codes = weight.to(torch.int8).contiguous()  # the kernel keeps the low `bits` bits of each byte
        packed = torch.empty((n, k * self.bits // 8), dtype=torch.int8, device=weight.device)
        check(lib().dcq_pack_nunchaku(C.c_void_p(codes.data_ptr()), n, k, self.bits, C.c_void_p(packed.data_ptr()),
                                      stream_of(weight.device)), "dcq_pack_nunchaku")
        return packed

    def pad_weight(self, weight: torch.Tensor) -> torch.Tensor:
        assert weight.ndim == 2, "weight tensor should be 2D."
        return pad(weight, divisor=(self.mem_n, self.mem_k * self.num_k_unrolls), dim=(0, 1))

    # ---- scales ------------------------------------------------------------------------------
    def check_if_micro_scale(self, group_size: int) -> bool:
        """Groups a quarter of an instruction's input depth wide carry float8 "micro" scales."""
        return group_size * 4 == self.insn_k

    def pad_scale(self, scale: torch.Tensor, group_size: int) -> torch.Tensor:
        """Channels up to a whole warp tile, groups up to what one pass consumes; new entries are 1."""
        grouped = group_size > 0 and scale.numel() > scale.shape[0]
        if not grouped:
            return pad(scale, divisor=self.warp_n, dim=0, fill_value=1)
        per_pass = self.insn_k // group_size if self.check_if_micro_scale(group_size) else self.num_k_unrolls
        return pad(scale.view(scale.shape[0], 1, -1, 1), divisor=(self.warp_n, per_pass), dim=(0, 2), fill_value=1)

    def pack_scale(self, scale: torch.Tensor, group_size: int) -> torch.Tensor:
        """16-bit scales [channels][groups] -> [groups][channels] with the channels of each 128-tile dealt out as
        (8 lanes, 4 lanes, 2 registers, 2 halves) instead of (8, 2, 4, 2): a lane's four scales end up adjacent."""
        if self.check_if_micro_scale(group_size):
            return self.pack_micro_scale(scale, group_size=group_size)
        assert scale.dtype in _HALF_TYPES, "currently nunchaku only supports fp16 and bf16."
        per_lane = min(max(self.warp_n // self.LANES, 2), 8)
        lanes = min(self.LANES, self.warp_n // per_lane)
        packs = self.warp_n // (per_lane * lanes)
        assert packs * lanes * per_lane == self.warp_n, "warp_n for scales should be equal to warp_n for weights."
        channels = scale.shape[0]
        tiles = scale.reshape(channels // self.warp_n, packs, lanes // 4, per_lane // 2, 4, 2, -1)
        dealt = tiles.permute(0, 6, 1, 2, 4, 3, 5).contiguous()  # tile, group, pack, lane / 4, lane % 4, register, half
        return dealt.view(-1) if group_size == -1 else dealt.view(-1, channels)

    def pack_micro_scale(self, scale: torch.Tensor, group_size: int) -> torch.Tensor:
        """float8 (e4m3) scales of 16-wide groups: four groups per instruction stay adjacent; the 128 channels of a
        tile are dealt out as (8 lanes, 4 lanes, 4 registers) instead of (4, 4, 8)."""
        assert scale.dtype in _HALF_TYPES, "currently nunchaku only supports fp16 and bf16."
        assert scale.max() <= 448, "scale should be less than 448."
        assert scale.min() >= -448, "scale should be greater than -448."
        assert group_size == 16, "currently only support group size 16."
        assert self.insn_k == 64, "insn_k should be 64."
        assert self.warp_n >= 32, "currently only support warp_n >= 32."
        per_lane = min(max(self.warp_n // self.LANES, 1), 4)
        packs = ceil_divide(self.warp_n, per_lane * self.LANES)
        assert packs * self.LANES * per_lane == self.warp_n, "warp_n for scales should be equal to warp_n for weights."
        channels = scale.shape[0]
        tiles = scale.to(torch.float8_e4m3fn).view(channels // self.warp_n, packs, per_lane, 4, 8, -1,
                                                   self.insn_k // group_size)
        return tiles.permute(0, 5, 1, 4, 3, 2, 6).contiguous().view(-1, channels)

    # ---- low-rank factors ---------------------------------------------------------------------
    def _lowrank_tile(self) -> tuple[int, int]:
        """(rows, columns) of one 16-bit fragment: 16 x 16 (two 16-bit values per register along the columns)."""
        return self.n_pack_size * self.N_LANES, self.k_pack_size * self.K_LANES * 2

    def pad_lowrank_weight(self, weight: torch.Tensor, down: bool) -> torch.Tensor:
        assert weight.ndim == 2, "weight tensor should be 2D."
        return pad(weight, divisor=self.warp_n, dim=1 if down else 0)

    def pack_lowrank_weight(self, weight: torch.Tensor, down: bool) -> torch.Tensor:
        """[rank][in] (down) or [out][rank] (up) -> the same number of elements as a grid of 16 x 16 fragments, each
        stored lane by lane (8 x 4 lanes of 2 x 2 registers of 2 values); the down projection's grid is transposed so
        the channel fragments of one rank fragment are adjacent."""
        assert weight.dtype in _HALF_TYPES, f"Unsupported weight dtype {weight.dtype}."
        tile_r, tile_c = self._lowrank_tile()
        weight = pad(weight, divisor=(tile_r, tile_c), dim=(0, 1))
        rows, cols = weight.shape
        grid = weight.view(rows // tile_r, tile_r, cols // tile_c, tile_c)
        grid = grid.permute(2, 0, 1, 3) if down else grid.permute(0, 2, 1, 3)
        shape = (cols, rows) if down else (rows, cols)
        lanes = grid.reshape(grid.shape[0], grid.shape[1], self.n_pack_size, self.N_LANES, 1, self.k_pack_size,
                             self.K_LANES, 2)
        return lanes.permute(0, 1, 3, 6, 2, 5, 4, 7).contiguous().view(shape)

    def unpack_lowrank_weight(self, weight: torch.Tensor, down: bool) -> torch.Tensor:
        """Inverse of ``pack_lowrank_weight`` (on the padded shape)."""
        assert weight.dtype in _HALF_TYPES, f"Unsupported weight dtype {weight.dtype}."
        tile_r, tile_c = self._lowrank_tile()
        c, r = weight.shape
        outer, inner = (c // tile_c, r // tile_r) if down else (c // tile_r, r // tile_c)
        lanes = weight.view(outer, inner, self.N_LANES, self.K_LANES, self.n_pack_size, self.k_pack_size, 1, 2)
        grid = lanes.permute(0, 1, 4, 2, 6, 5, 3, 7).contiguous().view(outer, inner, tile_r, tile_c)
        if down:
            return grid.permute(1, 2, 0, 3).contiguous().view(r, c)
        return grid.permute(0, 2, 1, 3).contiguous().view(c, r)


def _grouped_view(t: torch.Tensor, rows: int, cols: int, what: str) -> tuple[int, int]:
    """(groups, group size) of a [rows][1][groups][1] scale tensor over ``cols`` input channels."""
    assert t.ndim == 4, f"{what} tensor should be 4D."
    assert t.shape[1] == 1 and t.shape[3] == 1 and t.shape[0] == rows
    groups = t.shape[2]
    assert cols % groups == 0, f"input channel size should be a whole number of {what} groups."
    return groups, cols // groups


def convert_to_nunchaku_w4x4y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor,
                                              bias: torch.Tensor | None = None, smooth: torch.Tensor | None = None,
                                              lora: tuple[torch.Tensor, torch.Tensor] | None = None,
                                              float_point: bool = False, subscale: torch.Tensor | None = None):
    """(weight, scale, bias, smooth, lora, subscale) packed for Nunchaku's W4A4 linear, with the results of
    backend/nunchaku/utils.py:258-330.

    Codes come back from the dequantized 16-bit weight as ``w / scale [/ subscale]`` in float32, rounded to integers
    in [-8, 7] or mapped to the nearest float4 value's index; everything is padded to whole tiles and packed."""
    assert weight.ndim == 2, "weight tensor should be 2D."
    dtype, device = weight.dtype, weight.device
    assert dtype in _HALF_TYPES, "currently nunchaku only supports fp16 and bf16."
    assert scale is not None, "scale tensor is required for quantization."
    oc, ic = weight.shape
    per_tensor = scale.numel() == 1
    if per_tensor:
        scale = scale.reshape(1).expand(oc).reshape(oc, 1, 1, 1)
    groups, group_size = _grouped_view(scale, oc, ic, "scale")
    values = weight.to(torch.float32).view(oc, 1, groups, group_size) / scale.to(dtype=torch.float32, device=device)
    sub_size = group_size
    if subscale is not None:
        sub_groups, sub_size = _grouped_view(subscale, oc, ic, "subscale")
        assert group_size > sub_size and group_size % sub_size == 0, "group size should be divisible by subgroup size."
        values = values.view(oc, 1, sub_groups, sub_size) / subscale.to(dtype=torch.float32, device=device)
    values = values.view(oc, ic)
    if float_point:
        codes = fp_quantize(values)
        assert codes.min() >= 0 and codes.max() <= 15, "quantized weight should be in [0, 15]."
    else:
        codes = values.round_()
        assert codes.min() >= -8 and codes.max() <= 7, "quantized weight should be in [-8, 7]."
    packer = NunchakuWeightPacker(bits=4)

    def vector(t, length, fill):  # per-channel vectors travel as one-group "scales"
        t = torch.full([length, 1], fill, dtype=dtype, device=device) if t is None else t.view(-1, 1).to(dtype)
        return packer.pack_scale(packer.pad_scale(t, group_size=-1), group_size=-1)

    def grouped(t, size):
        return packer.pack_scale(packer.pad_scale(t.to(dtype), group_size=size), group_size=size if size < ic else -1)

    packed = packer.pack_weight(packer.pad_weight(codes.to(torch.int32)))
    packed_scale = grouped(scale, group_size)
    packed_sub = grouped(subscale, sub_size) if subscale is not None else None
    packed_bias, packed_smooth = vector(bias, oc, 0), vector(smooth, ic, 1)
    if lora is not None:
        lora = tuple(packer.pack_lowrank_weight(packer.pad_lowrank_weight(factor, down=down), down=down)
                     for factor, down in zip(lora, (True, False)))
    if per_tensor:
        packed_scale = packed_scale.view(-1)[:1].clone().view([1])
    return packed, packed_scale, packed_bias, packed_smooth, lora, packed_sub


def convert_to_nunchaku_w8x8y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor,
                                              bias: torch.Tensor | None = None):
    """(weight, scale, bias) packed for Nunchaku's W8A8 linear, with the results of utils.py:333-352: int8 codes
    ``round(w / scale)`` per output channel."""
    assert weight.ndim == 2, "weight tensor should be 2D."
    dtype, device = weight.dtype, weight.device
    assert dtype in _HALF_TYPES, "currently nunchaku only supports fp16 and bf16."
    assert scale is not None, "scale tensor is required for quantization."
    oc, ic = weight.shape
    scale = (scale.reshape(1).expand(oc) if scale.numel() == 1 else scale).reshape(oc, 1)
    codes = (weight.to(torch.float32) / scale.to(dtype=torch.float32, device=device)).round_().to(torch.int32)
    assert codes.min() >= -128 and codes.max() <= 127, "quantized weight should be in [-128, 127]."
    packer = NunchakuWeightPacker(bits=8)

    def vector(t):
        return packer.pack_scale(packer.pad_scale(t.to(dtype), group_size=-1), group_size=-1)

    bias = torch.zeros([oc, 1], dtype=dtype, device=device) if bias is None else bias.view(-1, 1)
    return packer.pack_weight(packer.pad_weight(codes.view(oc, ic))), vector(scale), vector(bias).view(-1)


def _scale_entry(scale: torch.Tensor, weight_ndim: int) -> tuple[torch.Tensor, str]:
    """A multi-element scale of a linear (or pointwise convolution) weight as [out][1][groups][1], and the engine's
    name for it: ``wcscales`` per output channel, ``wscales`` per group."""
    if scale.ndim > 2 * weight_ndim and scale.shape[2 * weight_ndim:].numel() == 1:
        # the quantizer's scale of a pointwise convolution weight: one more (1, 1) pair per kernel dimension
        scale = scale.view(scale.shape[:2 * weight_ndim])
    assert scale.ndim == 2 * weight_ndim, "one (count, size) pair of dimensions per weight dimension"
    rows, groups = scale.shape[0], scale.shape[2]
    assert scale.numel() == rows * groups, "scales vary along output channels and input groups only"
    return scale.view(rows, 1, groups, 1), "wcscales" if groups == 1 else "wscales"


def convert_to_nunchaku_w4x4y16_linear_state_dict(weight: torch.Tensor, scale: torch.Tensor,
                                                  bias: torch.Tensor | None = None, smooth: torch.Tensor | None = None,
                                                  lora: tuple[torch.Tensor, torch.Tensor] | None = None,
                                                  shift: torch.Tensor | None = None, smooth_fused: bool = False,
                                                  float_point: bool = False,
                                                  subscale: torch.Tensor | None = None) -> dict[str, torch.Tensor]:
    """One linear's Nunchaku tensors under the engine's key names, with the results of
    backend/nunchaku/convert.py:12-72.

    Before packing, the low-rank down projection is taken back to un-smoothed inputs (divided by ``smooth`` unless the
    smoothing is fused into the previous layer), and a constant input ``shift`` contributes ``up @ down @ shift``
    through the branch, which joins the bias (both in float64)."""
    if weight.ndim > 2:  # pointwise convolution: trailing dimensions of size 1
        assert weight.numel() == weight.shape[0] * weight.shape[1]
        weight = weight.view(weight.shape[0], weight.shape[1])
    weight_ndim = weight.ndim  # of the flattened weight, as the reference checks it (convert.py:24-30, 37)
    scale_key = "wtscale"
    if scale.numel() > 1:
        scale, scale_key = _scale_entry(scale, weight_ndim)
    subscale_key = None
    if subscale is not None:
        assert subscale.numel() > 1
        subscale, subscale_key = _scale_entry(subscale, weight_ndim)
    if lora is not None and not (smooth is None and shift is None):
        down, up = lora
        down64 = down.to(torch.float64)
        if smooth is not None and not smooth_fused:
            down64 = down64 / smooth.to(torch.float64)[None, :]
        if shift is not None:
            column = shift.to(torch.float64).reshape(-1, 1)
            if column.numel() == 1:
                column = column.expand(down64.shape[1], 1)
            through_branch = (up.to(torch.float64) @ down64 @ column).view(-1)
            base = bias.to(torch.float64) if bias is not None else torch.zeros_like(through_branch)
            bias = (base + through_branch).to(weight.dtype)
        lora = (down64.to(weight.dtype), up)
    packed, packed_scale, packed_bias, packed_smooth, packed_lora, packed_sub = convert_to_nunchaku_w4x4y16_linear_weight(
        weight, scale=scale, bias=bias, smooth=smooth, lora=lora, float_point=float_point, subscale=subscale)
    entries = [("qweight", packed), (scale_key, packed_scale)]
    if packed_sub is not None:
        entries.append((subscale_key, packed_sub))
    entries += [("bias", packed_bias), ("smooth_orig", packed_smooth),
                ("smooth", torch.ones_like(packed_smooth) if smooth_fused else packed_smooth.clone())]
    if packed_lora is not None:
        entries += [("lora_down", packed_lora[0]), ("lora_up", packed_lora[1])]
    return dict(entries)
