"""QServe checkpoint conversion (reference: deepcompressor/backend/qserve/utils.py:10-210,
backend/qserve/convert.py:14-146).

``scale.pt`` + ``model.pt`` of the weight stage -> QServe's W4A8 / W8A8 linear tensors.  The converter rebuilds the
integer codes from the dequantized float16 weights and the saved scales with the reference's own arithmetic (so it
only works on weights that are exact code * scale products); the regrouping of the 4-bit codes into the back end's
fragment order and their nibble packing is one HIP kernel (``dcq_pack_qserve_w4``).
"""

from __future__ import annotations

import ctypes as C

import torch

from .._lib import DcqError, check, lib, stream_of

__all__ = ["QServePacker", "convert_to_qserve_w4x8y16_linear_weight", "convert_to_qserve_w8x8y16_linear_weight",
           "convert_to_qserve_state_dict"]


class QServePacker:
    """backend/qserve/utils.py:10-62 with the geometry of ``MmaWeightPackerBase(bits=8, warp_n=32)``: tiles of 32
    output x 32 input channels."""

    mem_n, mem_k = 32, 32
    num_n_packs, n_pack_size, num_n_lanes, reg_n = 2, 2, 8, 1

    def pack_weight(self, weight: torch.Tensor) -> torch.Tensor:
        assert weight.dtype == torch.uint8, f"quantized weight should be torch.uint8, but got {weight.dtype}."
        if not weight.is_cuda:
            raise DcqError("deepcompressor_amd runs on MI355X GPUs only (there is no CPU path)")
        assert weight.min() >= 0, "quantized weight should be non-negative."
        assert weight.max() <= 15, "quantized weight should be less than 16."
        n, k = weight.shape
        assert n % self.mem_n == 0, f"output channel size ({n}) should be divisible by mem_n ({self.mem_n})."
        assert k % self.mem_k == 0, f"input channel size ({k}) should be divisible by mem_k ({self.mem_k})."
        weight = weight.contiguous()
        out = torch.empty((n, k // 2), dtype=torch.int8, device=weight.device)
        check(lib().dcq_pack_qserve_w4(C.c_void_p(weight.data_ptr()), n, k, C.c_void_p(out.data_ptr()),
                                       stream_of(weight.device)), "dcq_pack_qserve_w4")
        return out

    def pack_scale(self, scale: torch.Tensor, zero: torch.Tensor | None = None, subscale: torch.Tensor | None = None):
        scale = scale.view(-1)
        n = scale.shape[0]
        if subscale is None:
            zero = zero.view(-1)
        else:
            assert subscale.dtype == torch.int8, f"subscale should be torch.int8, but got {subscale.dtype}."
            view = (n // self.mem_n, self.num_n_packs, self.n_pack_size, self.num_n_lanes, self.reg_n, -1)
            subscale = subscale.view(view).permute(5, 0, 1, 3, 2, 4).contiguous().view(-1, n)
            zero = zero.view(view).permute(5, 0, 1, 3, 2, 4).contiguous().view(-1, n)
        return scale, zero, subscale


def convert_to_qserve_w4x8y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor,
                                            subscale: torch.Tensor | None = None, zero_pre_scaled: bool = False):
    """(packed weight int8 [oc][ic / 2], scale, zero, subscale) -- backend/qserve/utils.py:64-179."""
    dtype = weight.dtype
    assert dtype == torch.float16, "currently qserve only supports fp16."
    assert scale is not None, "scale tensor is required for quantization."
    assert zero is not None, "zero point tensor is required for quantization."
    weight = weight.to(dtype=torch.float32)
    scale = scale.to(dtype=torch.float32, device=weight.device)
    zero = zero.to(dtype=torch.float32, device=weight.device)
    oc, ic = weight.shape
    if scale.numel() == 1:
        scale = scale.view(-1).expand(oc)
    scale = scale.reshape(oc).contiguous().view(oc, 1)
    if subscale is not None:  # per-group quantization: int8 first level, 4-bit second level
        subscale = subscale.to(dtype=weight.dtype, device=weight.device)
        assert subscale.numel() > 1, "subscale tensor is required for per-group quantization."
        subscale = subscale.view(oc, -1, 1).round_()
        ng = subscale.shape[1]
        gs = ic // ng
        assert ic == ng * gs, "input channel size should be divisible by group size."
        if zero.numel() == 1:
            zero = zero.view(1, 1).expand(oc, ng)
        zero = zero.reshape(oc, ng).contiguous().view(oc, ng, 1).round_()
        weight = weight.div_(scale).round_()
        assert weight.min() >= -128, "first-level quantized weight should be greater than or equal to -128."
        assert weight.max() <= 127, "first-level quantized weight should be less than or equal to 127."
        weight = weight.view(oc, ng, gs)
        if not zero_pre_scaled:  # zero point is int8
            weight = weight.add_(zero)
        weight = weight.div_(subscale)
        if zero_pre_scaled:  # zero point is int4
            if zero.min() < 0:  # sint4 zero point
                zero = zero.add_(8)
            assert zero.min() >= 0, "quantized zero point should be non-negative."
            assert zero.max() <= 15, "quantized zero point should be less than 16."
            weight = weight.add_(zero)
            zero = zero.mul_(subscale)
        elif weight.min() < 0:  # sint4 weight
            weight = weight.add_(8)
            zero = zero.add_(8 * subscale)
        first_level = weight.mul(subscale)
        assert first_level.min() >= 0, "first-level dequantize weight should be non-negative."
        assert first_level.max() <= 255, "first-level dequantize weight should be less than 256."
        del first_level
        assert subscale.min() >= 0, "subscale should be non-negative."
        assert subscale.max() <= 127, "subscale should be less than or equal to 127."
        assert zero.min() >= 0, "quantized zero point should be non-negative."
        assert zero.max() <= 255, "quantized zero point should be less than 256."
        assert weight.min() >= 0, "quantized weight should be non-negative."
        assert weight.max() <= 15, "quantized weight should be less than 16."
        zero = -zero  # for group quantization QServe uses q * s + z = r instead of q * s - z = r
        subscale = subscale.to(torch.int8)
        zero = zero.to(torch.int8)
    else:  # per-channel quantization
        if zero.numel() == 1:
            zero = zero.view(-1).expand(oc)
        zero = zero.reshape(oc).contiguous().view(oc, 1)
        if not zero_pre_scaled:  # zero point is fp16
            weight = weight.add_(zero)
        weight = weight.div_(scale).round_()
        if zero_pre_scaled:  # zero point is int4
            zero = zero.round_()
            if zero.min() < 0:
                zero = zero.add_(8)
            assert zero.min() >= 0, "quantized zero point should be non-negative."
            assert zero.max() <= 15, "quantized zero point should be less than 16."
            weight = weight.add_(zero)
            zero = zero.mul_(scale)
        elif weight.min() < 0:
            weight = weight.add_(8)
            zero = zero.add_(8 * scale)
        assert weight.min() >= 0, "quantized weight should be non-negative."
        assert weight.max() <= 15, "quantized weight should be less than 16."
        zero = zero.to(dtype=dtype)
    scale = scale.to(dtype=dtype)
    packer = QServePacker()
    weight = packer.pack_weight(weight.view(oc, ic).to(torch.uint8))
    scale, zero, subscale = packer.pack_scale(scale=scale, zero=zero, subscale=subscale)
    return weight, scale, zero, subscale


def convert_to_qserve_w8x8y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor):
    """(int8 weight, float16 per-channel scale) -- backend/qserve/utils.py:182-210."""
    dtype = weight.dtype
    assert dtype == torch.float16, "currently qserve only supports fp16."
    assert scale is not None, "scale tensor is required for quantization."
    weight = weight.to(dtype=torch.float32)
    scale = scale.to(dtype=torch.float32, device=weight.device)
    oc = weight.shape[0]
    if scale.numel() == 1:
        scale = scale.view(-1).expand(oc)
    scale = scale.reshape(oc).contiguous().view(oc, 1)
    weight = weight.div_(scale).round_()
    assert weight.min() >= -128, "quantized weight should be greater than or equal to -128."
    assert weight.max() <= 127, "quantized weight should be less than or equal to 127."
    return weight.contiguous().to(torch.int8), scale.view(oc).to(dtype=dtype)


def convert_to_qserve_state_dict(state_dict: dict[str, torch.Tensor], scale_dict: dict[str, torch.Tensor],
                                 weight_bits: int, device: torch.device | str = "cuda") -> dict[str, torch.Tensor]:
    """``model.pt`` + ``scale.pt`` -> QServe tensors (backend/qserve/convert.py:83-146): per quantized linear
    ``qweight / s1_scales / s1_szeros`` (per channel) or ``qweight / s1_scales / s2_scales / s2_zeros`` (per group)
    for 4 bits, ``weight / dequant_scale`` for 8 bits; everything else is copied."""
    assert weight_bits in (4, 8), "weight bits should be 4 or 8."
    scales: dict[str, dict[tuple[int, ...], torch.Tensor]] = {}
    zeros: dict[str, tuple[torch.Tensor | None, bool]] = {}
    for name, tensor in scale_dict.items():
        if name.endswith("zero"):
            zero = None if tensor is None or bool((tensor == 0).all()) else tensor
            if name.endswith(".scaled_zero"):
                zeros[name[:-12]] = (zero, False)
            else:
                zeros[name[:-5]] = (zero, True)
        else:
            assert ".weight.scale" in name
            idx = name.index(".weight.scale")
            level = tuple(map(int, name[idx + 14:].split(".")))
            scales.setdefault(name[: idx + 7], {})[level] = tensor
    for param_name in zeros:
        assert param_name in state_dict, f"zero point tensor {param_name} not found in state dict."
        assert param_name in scales, f"scale tensor {param_name} not found in scale dict."
    converted: dict[str, torch.Tensor] = {}
    for param_name, param in state_dict.items():
        if param_name not in scales:
            converted[param_name] = param.clone().cpu() if isinstance(param, torch.Tensor) else param
            continue
        module_name = param_name[:-7]
        weight = param.data.to(device).clone()
        zero, pre_scaled = zeros.get(param_name, (None, False))
        levels = sorted(scales[param_name].items(), key=lambda kv: kv[0])
        assert len(levels) <= 2, "more than two scale levels are not supported."
        scale = levels[0][1].clone()
        subscale = levels[1][1].clone() if len(levels) > 1 else None
        if weight_bits == 4:
            zero = zero.clone() if zero is not None else None
            w, s, z, ss = convert_to_qserve_w4x8y16_linear_weight(weight, scale=scale, zero=zero, subscale=subscale,
                                                                  zero_pre_scaled=pre_scaled)
            converted[f"{module_name}.qweight"], converted[f"{module_name}.s1_scales"] = w.cpu(), s.cpu()
            if ss is None:
                converted[f"{module_name}.s1_szeros"] = z.cpu()
            else:
                converted[f"{module_name}.s2_scales"], converted[f"{module_name}.s2_zeros"] = ss.cpu(), z.cpu()
        else:
            assert zero is None, "zero point tensor is not supported for W8 quantization."
            assert subscale is None, "subscale tensor is not supported for W8 quantization."
            w, s = convert_to_qserve_w8x8y16_linear_weight(weight, scale=scale)
            converted[f"{module_name}.weight"], converted[f"{module_name}.dequant_scale"] = w.cpu(), s.cpu()
    return converted
