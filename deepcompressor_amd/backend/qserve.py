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
        """Per-channel values flat; the per-group int8 ``subscale`` / ``zero`` tables transposed to [groups][channels]
        with the 32 channels of a tile dealt as (2 packs, 8 lanes, 2 registers) instead of (2, 2, 8)."""
        channels = scale.numel()
        if subscale is None:
            return scale.view(-1), zero.view(-1), None
        assert subscale.dtype == torch.int8, f"subscale should be torch.int8, but got {subscale.dtype}."

        def deal(table: torch.Tensor) -> torch.Tensor:
            tiles = table.view(channels // self.mem_n, self.num_n_packs, self.n_pack_size, self.num_n_lanes,
                               self.reg_n, -1)
            return tiles.permute(5, 0, 1, 3, 2, 4).contiguous().view(-1, channels)

        return scale.view(-1), deal(zero), deal(subscale)


def _in_range(t: torch.Tensor, lo: float, hi: float, what: str) -> None:
    """Converter invariant: every value of ``t`` lies in [lo, hi] (an AssertionError naming ``what`` otherwise)."""
    vmin, vmax = float(t.min()), float(t.max())
    assert lo <= vmin and vmax <= hi, f"{what} should lie in [{lo:g}, {hi:g}], got [{vmin:g}, {vmax:g}]."


def _per_row(t: torch.Tensor, rows: int, cols: int = 1) -> torch.Tensor:
    """A scalar, per-row or per-(row, group) parameter as a float32 [rows][cols][1] (cols == 1: [rows][1]) tensor."""
    t = t.reshape(-1).expand(rows * cols) if t.numel() == 1 else t
    t = t.reshape(rows, cols).contiguous()
    return t.view(rows, cols, 1) if cols > 1 else t.view(rows, 1)


def _unsigned4(codes: torch.Tensor, zero: torch.Tensor, step: torch.Tensor, zero_is_code: bool):
    """Bring 4-bit codes and their zero point to the unsigned convention QServe stores.

    ``zero_is_code`` (the checkpoint's PreScale zero): the zero point is itself a 4-bit code (signed ones are shifted
    by 8), it is added to the codes and then expressed in units of the first-level grid (times ``step``).  Otherwise
    the zero point already lives on that grid; signed codes are shifted by 8 and the zero point by 8 steps."""
    if zero_is_code:
        zero = zero + 8 if zero.min() < 0 else zero
        _in_range(zero, 0, 15, "quantized zero point")
        return codes + zero, zero * step
    if codes.min() < 0:
        return codes + 8, zero + 8 * step
    return codes, zero


def convert_to_qserve_w4x8y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor,
                                            subscale: torch.Tensor | None = None, zero_pre_scaled: bool = False):
    """(packed weight int8 [oc][ic / 2], scale, zero, subscale) with the results of backend/qserve/utils.py:64-179.

    The 4-bit codes are recovered from the dequantized float16 weight: per channel ``round((w [+ z]) / s)``; per group
    first the 8-bit level ``round(w / s)`` (checked against [-128, 127]), then ``(that [+ z]) / subscale`` inside each
    group.  The order of the float32 operations is the reference's, so exact code * scale products come back as the
    same integers."""
    dtype = weight.dtype
    assert dtype == torch.float16, "currently qserve only supports fp16."
    assert scale is not None, "scale tensor is required for quantization."
    assert zero is not None, "zero point tensor is required for quantization."
    oc, ic = weight.shape
    dev = weight.device
    w = weight.to(torch.float32)
    s1 = _per_row(scale.to(dtype=torch.float32, device=dev), oc)
    z = zero.to(dtype=torch.float32, device=dev)
    if subscale is None:  # one float16 scale per output channel
        z = _per_row(z, oc)
        if zero_pre_scaled:
            codes = (w / s1).round_()
            codes, z = _unsigned4(codes, z.round(), s1, True)
        else:
            codes = ((w + z) / s1).round_()
            codes, z = _unsigned4(codes, z, s1, False)
        _in_range(codes, 0, 15, "quantized weight")
        z = z.to(dtype)
    else:  # int8 first level per channel, unsigned 4-bit second level per group
        assert subscale.numel() > 1, "subscale tensor is required for per-group quantization."
        s2 = subscale.to(dtype=torch.float32, device=dev).view(oc, -1, 1).round()
        groups = s2.shape[1]
        assert ic % groups == 0, "input channel size should be divisible by group size."
        z = _per_row(z, oc, groups).view(oc, groups, 1).round()
        level1 = (w / s1).round_()
        _in_range(level1, -128, 127, "first-level quantized weight")
        level1 = level1.view(oc, groups, ic // groups)
        if zero_pre_scaled:
            codes, z = _unsigned4(level1 / s2, z, s2, True)
        else:
            codes, z = _unsigned4((level1 + z) / s2, z, s2, False)
        _in_range(codes * s2, 0, 255, "first-level dequantize weight")
        _in_range(s2, 0, 127, "subscale")
        _in_range(z, 0, 255, "quantized zero point")
        _in_range(codes, 0, 15, "quantized weight")
        z = (-z).to(torch.int8)  # QServe's per-group convention is q * s + z
        subscale = s2.to(torch.int8)
    packer = QServePacker()
    packed = packer.pack_weight(codes.reshape(oc, ic).to(torch.uint8))
    out_scale, out_zero, out_subscale = packer.pack_scale(scale=s1.to(dtype), zero=z, subscale=subscale)
    return packed, out_scale, out_zero, out_subscale


def convert_to_qserve_w8x8y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor):
    """(int8 weight, float16 per-channel scale) with the results of backend/qserve/utils.py:182-210."""
    assert weight.dtype == torch.float16, "currently qserve only supports fp16."
    assert scale is not None, "scale tensor is required for quantization."
    oc = weight.shape[0]
    s = _per_row(scale.to(dtype=torch.float32, device=weight.device), oc)
    codes = (weight.to(torch.float32) / s).round_()
    _in_range(codes, -128, 127, "quantized weight")
    return codes.contiguous().to(torch.int8), s.view(oc).to(torch.float16)


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
