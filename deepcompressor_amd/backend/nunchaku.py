"""Nunchaku (SVDQuant inference engine) weight conversion
(reference: deepcompressor/backend/nunchaku/utils.py:16-368, backend/utils.py:12-58, 81-157).

W4A4 (integer or float4 codes, per-group scales, optional two-level micro scales, low-rank branch) and W8A8 linear
tensors in the back end's fragment order.  The code regrouping + bit packing is one HIP kernel
(``dcq_pack_nunchaku``); the scale / bias / low-rank permutations are small tensor reshapes.
"""

from __future__ import annotations

import ctypes as C
import typing as tp

import torch

from .._lib import DcqError, check, lib, stream_of
from .tinychat import convert_to_tinychat_w4x16y16_linear_weight

__all__ = ["ceil_divide", "pad", "fp_quantize", "NunchakuWeightPacker", "convert_to_nunchaku_w4x4y16_linear_weight",
           "convert_to_nunchaku_w8x8y16_linear_weight", "convert_to_nunchaku_w4x4y16_linear_state_dict", "convert_to_nunchaku_w4x16_linear_weight"]


def ceil_divide(x: int, divisor: int) -> int:
    return (x + divisor - 1) // divisor


def pad(tensor: torch.Tensor | None, divisor: int | tp.Sequence[int], dim: int | tp.Sequence[int],
        fill_value: float | int = 0) -> torch.Tensor | None:
    """Grow the given dimensions to multiples of ``divisor`` (backend/utils.py:29-58)."""
    if isinstance(divisor, int):
        if divisor <= 1:
            return tensor
    elif all(d <= 1 for d in divisor):
        return tensor
    if tensor is None:
        return None
    shape = list(tensor.shape)
    if isinstance(dim, int):
        assert isinstance(divisor, int)
        shape[dim] = ceil_divide(shape[dim], divisor) * divisor
    else:
        if isinstance(divisor, int):
            divisor = [divisor] * len(dim)
        for d, div in zip(dim, divisor, strict=True):
            shape[d] = ceil_divide(shape[d], div) * div
    if shape == list(tensor.shape):
        return tensor.clone()
    result = torch.full(shape, fill_value, dtype=tensor.dtype, device=tensor.device)
    result[tuple(slice(0, extent) for extent in tensor.shape)] = tensor
    return result


_FP4 = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0]


def fp_quantize(x: torch.Tensor, codebook: torch.Tensor | None = None) -> torch.Tensor:
    """Index of the nearest codebook entry, first one on ties (backend/utils.py:81-88); row blocks bound the
    broadcast difference tensor."""
    if codebook is None:
        codebook = torch.tensor(_FP4, dtype=x.dtype, device=x.device)
    flat = x.reshape(-1)
    out = torch.empty(flat.shape, dtype=torch.long, device=x.device)
    step = 1 << 22
    for i in range(0, flat.numel(), step):
        out[i:i + step] = (flat[i:i + step].unsqueeze(-1) - codebook.unsqueeze(0)).abs().argmin(dim=-1)
    return out.view(x.shape)


class NunchakuWeightPacker:
    """backend/nunchaku/utils.py:16-255 with the geometry of ``MmaWeightPackerBase(bits, warp_n)``."""

    def __init__(self, bits: int, warp_n: int = 128):
        assert bits in (4, 8), f"weight bits {bits} is not supported."
        self.bits, self.warp_n = bits, warp_n
        self.comp_n, self.comp_k = 16, 256 // bits
        self.insn_n, self.insn_k = 8, self.comp_k
        self.num_lanes, self.num_k_lanes, self.num_n_lanes = 32, 4, 8
        assert warp_n >= self.comp_n and warp_n % self.comp_n == 0
        self.reg_k, self.reg_n = 32 // bits, 1
        self.k_pack_size = self.comp_k // (self.num_k_lanes * self.reg_k)
        self.n_pack_size = self.comp_n // (self.num_n_lanes * self.reg_n)
        self.mem_k, self.mem_n = self.comp_k, warp_n
        self.num_k_packs = self.mem_k // (self.k_pack_size * self.num_k_lanes * self.reg_k)
        self.num_n_packs = self.mem_n // (self.n_pack_size * self.num_n_lanes * self.reg_n)
        self.num_k_unrolls = 2

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        assert weight.dtype == torch.int32, f"quantized weight should be torch.int32, but got {weight.dtype}."
        if not weight.is_cuda:
            raise DcqError("deepcompressor_amd runs on MI355X GPUs only (there is no CPU path)")
        n, k = weight.shape
        assert n % self.mem_n == 0, f"output channel size ({n}) should be divisible by mem_n ({self.mem_n})."
        assert k % (self.mem_k * self.num_k_unrolls) == 0, (
            f"input channel size ({k}) should be divisible by mem_k ({self.mem_k}) * num_k_unrolls ({self.num_k_unrolls}).")
        assert self.warp_n == 128, "the pack kernel is built for warp_n = 128"
        codes = weight.to(torch.int8).contiguous()  # only the low bits survive the packing, as in the reference's mask
        out = torch.empty((n, k * self.bits // 8), dtype=torch.int8, device=weight.device)
        check(lib().dcq_pack_nunchaku(C.c_void_p(codes.data_ptr()), n, k, self.bits, C.c_void_p(out.data_ptr()),
                                      stream_of(weight.device)), "dcq_pack_nunchaku")
        return out

    def check_if_micro_scale(self, group_size: int) -> bool:
        return self.insn_k == group_size * 4

    def pack_scale(self, scale: torch.Tensor, group_size: int) -> torch.Tensor:
        if self.check_if_micro_scale(group_size=group_size):
            return self.pack_micro_scale(scale, group_size=group_size)
        assert scale.dtype in (torch.float16, torch.bfloat16), "currently nunchaku only supports fp16 and bf16."
        n = scale.shape[0]
        s_pack_size = min(max(self.warp_n // self.num_lanes, 2), 8)
        num_s_lanes = min(self.num_lanes, self.warp_n // s_pack_size)
        num_s_packs = self.warp_n // (s_pack_size * num_s_lanes)
        warp_s = num_s_packs * num_s_lanes * s_pack_size
        assert warp_s == self.warp_n, "warp_n for scales should be equal to warp_n for weights."
        scale = scale.reshape(n // warp_s, num_s_packs, num_s_lanes // 4, s_pack_size // 2, 4, 2, -1)
        scale = scale.permute(0, 6, 1, 2, 4, 3, 5).contiguous()
        return scale.view(-1) if group_size == -1 else scale.view(-1, n)

    def pack_micro_scale(self, scale: torch.Tensor, group_size: int) -> torch.Tensor:
        assert scale.dtype in (torch.float16, torch.bfloat16), "currently nunchaku only supports fp16 and bf16."
        assert scale.max() <= 448, "scale should be less than 448."
        assert scale.min() >= -448, "scale should be greater than -448."
        assert group_size == 16, "currently only support group size 16."
        assert self.insn_k == 64, "insn_k should be 64."
        scale = scale.to(dtype=torch.float8_e4m3fn)
        n = scale.shape[0]
        assert self.warp_n >= 32, "currently only support warp_n >= 32."
        s_pack_size = min(max(self.warp_n // self.num_lanes, 1), 4)
        num_s_lanes = 4 * 8
        num_s_packs = ceil_divide(self.warp_n, s_pack_size * num_s_lanes)
        warp_s = num_s_packs * num_s_lanes * s_pack_size
        assert warp_s == self.warp_n, "warp_n for scales should be equal to warp_n for weights."
        scale = scale.view(n // warp_s, num_s_packs, s_pack_size, 4, 8, -1, self.insn_k // group_size)
        scale = scale.permute(0, 5, 1, 4, 3, 2, 6).contiguous()
        return scale.view(-1, n)

    def _lowrank_packs(self) -> tuple[int, int, int, int]:
        reg_n, reg_k = 1, 2  # 32 bits / 16 bits
        return reg_n, reg_k, self.n_pack_size * self.num_n_lanes * reg_n, self.k_pack_size * self.num_k_lanes * reg_k

    def pack_lowrank_weight(self, weight: torch.Tensor, down: bool) -> torch.Tensor:
        assert weight.dtype in (torch.float16, torch.bfloat16), f"Unsupported weight dtype {weight.dtype}."
        reg_n, reg_k, pack_n, pack_k = self._lowrank_packs()
        weight = pad(weight, divisor=(pack_n, pack_k), dim=(0, 1))
        if down:
            r, c = weight.shape
            r_packs, c_packs = r // pack_n, c // pack_k
            weight = weight.view(r_packs, pack_n, c_packs, pack_k).permute(2, 0, 1, 3)
        else:
            c, r = weight.shape
            c_packs, r_packs = c // pack_n, r // pack_k
            weight = weight.view(c_packs, pack_n, r_packs, pack_k).permute(0, 2, 1, 3)
        weight = weight.reshape(c_packs, r_packs, self.n_pack_size, self.num_n_lanes, reg_n, self.k_pack_size,
                                self.num_k_lanes, reg_k)
        weight = weight.permute(0, 1, 3, 6, 2, 5, 4, 7).contiguous()
        return weight.view(c, r)

    def unpack_lowrank_weight(self, weight: torch.Tensor, down: bool) -> torch.Tensor:
        c, r = weight.shape
        assert weight.dtype in (torch.float16, torch.bfloat16), f"Unsupported weight dtype {weight.dtype}."
        reg_n, reg_k, pack_n, pack_k = self._lowrank_packs()
        if down:
            r_packs, c_packs = r // pack_n, c // pack_k
        else:
            c_packs, r_packs = c // pack_n, r // pack_k
        weight = weight.view(c_packs, r_packs, self.num_n_lanes, self.num_k_lanes, self.n_pack_size, self.k_pack_size,
                             reg_n, reg_k)
        weight = weight.permute(0, 1, 4, 2, 6, 5, 3, 7).contiguous().view(c_packs, r_packs, pack_n, pack_k)
        if down:
            return weight.permute(1, 2, 0, 3).contiguous().view(r, c)
        return weight.permute(0, 2, 1, 3).contiguous().view(c, r)

    def pad_weight(self, weight: torch.Tensor) -> torch.Tensor:
        assert weight.ndim == 2, "weight tensor should be 2D."
        return pad(weight, divisor=(self.mem_n, self.mem_k * self.num_k_unrolls), dim=(0, 1))

    def pad_scale(self, scale: torch.Tensor, group_size: int) -> torch.Tensor:
        if group_size > 0 and scale.numel() > scale.shape[0]:
            scale = scale.view(scale.shape[0], 1, -1, 1)
            if self.check_if_micro_scale(group_size=group_size):
                return pad(scale, divisor=(self.warp_n, self.insn_k // group_size), dim=(0, 2), fill_value=1)
            return pad(scale, divisor=(self.warp_n, self.num_k_unrolls), dim=(0, 2), fill_value=1)
        return pad(scale, divisor=self.warp_n, dim=0, fill_value=1)

    def pad_lowrank_weight(self, weight: torch.Tensor, down: bool) -> torch.Tensor:
        assert weight.ndim == 2, "weight tensor should be 2D."
        return pad(weight, divisor=self.warp_n, dim=1 if down else 0)


def convert_to_nunchaku_w4x4y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor,
                                              bias: torch.Tensor | None = None, smooth: torch.Tensor | None = None,
                                              lora: tuple[torch.Tensor, torch.Tensor] | None = None,
                                              float_point: bool = False, subscale: torch.Tensor | None = None):
    """(weight, scale, bias, smooth, lora, subscale) packed for Nunchaku's W4A4 linear -- utils.py:258-330."""
    assert weight.ndim == 2, "weight tensor should be 2D."
    device, dtype = weight.device, weight.dtype
    assert dtype in (torch.float16, torch.bfloat16), "currently nunchaku only supports fp16 and bf16."
    assert scale is not None, "scale tensor is required for quantization."
    oc, ic = weight.shape
    per_tensor_scale = scale.numel() == 1
    if per_tensor_scale:
        scale = scale.view(-1).expand(oc).reshape(oc, 1, 1, 1)
    assert scale.ndim == 4, "scale tensor should be 4D."
    assert scale.shape[1] == scale.shape[3] == 1
    assert scale.shape[0] == oc
    ng, gs = scale.shape[2], ic // scale.shape[2]
    assert ic == gs * ng, "input channel size should be equal to group size times number of groups."
    if subscale is not None:
        assert subscale.ndim == 4, "subscale tensor should be 4D."
        assert subscale.shape[1] == subscale.shape[3] == 1
        assert subscale.shape[0] == oc
        nsg, sgs = subscale.shape[2], ic // subscale.shape[2]
        assert ic == sgs * nsg, "input channel size should be equal to subgroup size times number of subgroups."
        assert gs > sgs and gs % sgs == 0, "group size should be divisible by subgroup size."
    else:
        nsg, sgs = ng, gs
    weight = weight.to(dtype=torch.float32).view(oc, 1, ng, gs).div_(scale.to(dtype=torch.float32, device=device))
    if subscale is not None:
        weight = weight.view(oc, 1, nsg, sgs).div_(subscale.to(dtype=torch.float32, device=device))
    weight = weight.view(oc, ic)
    if float_point:
        weight = fp_quantize(weight)
        assert weight.min() >= 0 and weight.max() <= 15, "quantized weight should be in [0, 15]."
    else:
        weight = weight.round_()
        assert weight.min() >= -8 and weight.max() <= 7, "quantized weight should be in [-8, 7]."
    bias = torch.zeros([oc, 1], dtype=dtype, device=device) if bias is None else bias.view(-1, 1)
    smooth = torch.ones([ic, 1], dtype=dtype, device=device) if smooth is None else smooth.view(-1, 1)
    packer = NunchakuWeightPacker(bits=4)
    weight = packer.pad_weight(weight.to(dtype=torch.int32))
    scale = packer.pad_scale(scale.to(dtype=dtype), group_size=gs)
    if subscale is not None:
        subscale = packer.pad_scale(subscale.to(dtype=dtype), group_size=sgs)
    bias = packer.pad_scale(bias.to(dtype=dtype), group_size=-1)
    smooth = packer.pad_scale(smooth.to(dtype=dtype), group_size=-1)
    weight = packer.pack_weight(weight)
    scale = packer.pack_scale(scale, group_size=gs if gs < ic else -1)
    if subscale is not None:
        subscale = packer.pack_scale(subscale, group_size=sgs if sgs < ic else -1)
    bias = packer.pack_scale(bias, group_size=-1)
    smooth = packer.pack_scale(smooth, group_size=-1)
    if lora is not None:
        lora = (packer.pack_lowrank_weight(packer.pad_lowrank_weight(lora[0], down=True), down=True),
                packer.pack_lowrank_weight(packer.pad_lowrank_weight(lora[1], down=False), down=False))
    if per_tensor_scale:
        scale = scale.view(-1)[0].view([1])
    return weight, scale, bias, smooth, lora, subscale


def convert_to_nunchaku_w8x8y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor,
                                              bias: torch.Tensor | None = None):
    """(weight, scale, bias) packed for Nunchaku's W8A8 linear -- utils.py:333-352."""
    assert weight.ndim == 2, "weight tensor should be 2D."
    device, dtype = weight.device, weight.dtype
    assert dtype in (torch.float16, torch.bfloat16), "currently nunchaku only supports fp16 and bf16."
    assert scale is not None, "scale tensor is required for quantization."
    oc, ic = weight.shape
    if scale.numel() == 1:
        scale = scale.view(-1).expand(oc)
    scale = scale.reshape(oc, 1)
    weight = weight.to(dtype=torch.float32)
    weight = weight.div_(scale.to(dtype=torch.float32, device=device)).round_().to(torch.int32).view(oc, ic)
    assert weight.min() >= -128 and weight.max() <= 127, "quantized weight should be in [-128, 127]."
    bias = torch.zeros([oc, 1], dtype=dtype, device=device) if bias is None else bias.view(-1, 1)
    packer = NunchakuWeightPacker(bits=8)
    weight = packer.pack_weight(packer.pad_weight(weight))
    scale = packer.pack_scale(packer.pad_scale(scale.to(dtype=dtype), group_size=-1), group_size=-1)
    bias = packer.pack_scale(packer.pad_scale(bias.to(dtype=dtype), group_size=-1), group_size=-1).view(-1)
    return weight, scale, bias


def convert_to_nunchaku_w4x4y16_linear_state_dict(weight: torch.Tensor, scale: torch.Tensor,
                                                  bias: torch.Tensor | None = None, smooth: torch.Tensor | None = None,
                                                  lora: tuple[torch.Tensor, torch.Tensor] | None = None,
                                                  shift: torch.Tensor | None = None, smooth_fused: bool = False,
                                                  float_point: bool = False,
                                                  subscale: torch.Tensor | None = None) -> dict[str, torch.Tensor]:
    """One linear's Nunchaku tensors under the engine's key names (backend/nunchaku/convert.py:12-72): the low-rank
    down projection is un-smoothed, and a shifted input's constant contribution through the branch joins the bias."""
    if weight.ndim > 2:  # pointwise convolution
        assert weight.numel() == weight.shape[0] * weight.shape[1]
        weight = weight.view(weight.shape[0], weight.shape[1])
    if scale.numel() > 1:
        assert scale.ndim == weight.ndim * 2
        assert scale.numel() == scale.shape[0] * scale.shape[2]
        scale = scale.view(scale.shape[0], 1, scale.shape[2], 1)
        scale_key = "wcscales" if scale.shape[2] == 1 else "wscales"
    else:
        scale_key = "wtscale"
    subscale_key = ""
    if subscale is not None:
        assert subscale.ndim == weight.ndim * 2
        assert subscale.numel() == subscale.shape[0] * subscale.shape[2]
        assert subscale.numel() > 1
        subscale = subscale.view(subscale.shape[0], 1, subscale.shape[2], 1)
        subscale_key = "wcscales" if subscale.shape[2] == 1 else "wscales"
    if lora is not None and (smooth is not None or shift is not None):
        dtype = weight.dtype
        lora_down, lora_up = lora
        lora_down = lora_down.to(dtype=torch.float64)
        if smooth is not None and not smooth_fused:
            lora_down = lora_down.div_(smooth.to(torch.float64).unsqueeze(0))
        if shift is not None:
            bias = (torch.zeros([lora_up.shape[0]], dtype=torch.float64, device=weight.device) if bias is None
                    else bias.to(torch.float64))
            if shift.numel() == 1:
                shift = shift.view(1, 1).expand(lora_down.shape[1], 1).to(torch.float64)
            else:
                shift = shift.view(-1, 1).to(torch.float64)
            bias = bias.add_((lora_up.to(dtype=torch.float64) @ lora_down @ shift).view(-1)).to(dtype=dtype)
        lora = (lora_down.to(dtype=dtype), lora_up)
    weight, scale, bias, smooth, lora, subscale = convert_to_nunchaku_w4x4y16_linear_weight(
        weight, scale=scale, bias=bias, smooth=smooth, lora=lora, float_point=float_point, subscale=subscale)
    out: dict[str, torch.Tensor] = {"qweight": weight, scale_key: scale}
    if subscale is not None:
        out[subscale_key] = subscale
    out["bias"], out["smooth_orig"] = bias, smooth
    out["smooth"] = torch.ones_like(smooth) if smooth_fused else smooth.clone()
    if lora is not None:
        out["lora_down"], out["lora_up"] = lora
    return out


def convert_to_nunchaku_w4x16_linear_weight(weight: torch.Tensor, scale: torch.Tensor,
                                            zero: torch.Tensor | None = None, bias: torch.Tensor | None = None,
                                            adanorm_splits: int = 1):
    """(weight int32, scale, zero, bias) of a W4A16 linear in the TinyChat layout, as Nunchaku's AdaNorm layers take
    it (nunchaku/utils.py:341-368): the output channels of the ``adanorm_splits`` chunks are interleaved, and the
    scale chunks' bias gets the "1 +" of the modulation folded in."""
    oc, ic = weight.shape
    assert scale.ndim == 4, "scale tensor should be 4D."
    assert scale.shape[0] == oc
    assert scale.shape[1] == scale.shape[3] == 1
    ng = scale.shape[2]
    if bias is None:
        bias = torch.zeros([oc], dtype=weight.dtype, device=weight.device)
    assert oc % adanorm_splits == 0, "output channel size should be divisible by splits."
    if adanorm_splits > 1:
        weight = weight.view(adanorm_splits, oc // adanorm_splits, ic).transpose(0, 1).reshape(oc, ic)
        scale = scale.view(adanorm_splits, oc // adanorm_splits, ng).transpose(0, 1).reshape(oc, 1, ng, 1)
        bias = bias.reshape(adanorm_splits, oc // adanorm_splits).transpose(0, 1)
        delta = [0] * adanorm_splits
        delta[1] = delta[-2] = 1
        bias = bias.add_(torch.tensor(delta, dtype=bias.dtype, device=bias.device))
        bias = bias.reshape(oc)
    weight, scale, zero = convert_to_tinychat_w4x16y16_linear_weight(
        weight=weight, scale=scale, zero=torch.full_like(scale, 7) if zero is None else zero, zero_pre_scaled=True)
    return weight.view(torch.int32), scale, zero, bias
