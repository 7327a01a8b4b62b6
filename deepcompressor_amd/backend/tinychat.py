"""TinyChat (AWQ W4A16) weight conversion
(reference: deepcompressor/backend/tinychat/utils.py:11-107, backend/tinychat/convert.py:14-101).

The integer codes are recovered from the dequantized 16-bit weights and the stored scales / zeros exactly as the
reference does, then folded into the kernel's int16 layout by ``dcq_pack_tinychat_w4``.
"""

from __future__ import annotations

import ctypes as C

import torch

from .._lib import DcqError, check, lib, stream_of

__all__ = ["ceil_num_groups", "pack_w4", "convert_to_tinychat_w4x16y16_linear_weight",
           "convert_to_tinychat_w4x16y16_linear_state_dict", "convert_to_tinychat_state_dict"]


def _ceil_divide(x: int, d: int) -> int:
    return (x + d - 1) // d


def ceil_num_groups(in_features: int, group_size: int, weight_bits: int = 4) -> int:
    """Number of scale rows the TinyChat kernel expects (tinychat/utils.py:11-42): whole int32 packs of groups, and
    for groups narrower than 128 a multiple of 128 / group_size packs."""
    assert in_features % group_size == 0, "input channel size should be divisible by group size."
    num_groups = in_features // group_size
    assert weight_bits in (4, 2, 1), "weight bits should be 4, 2, or 1."
    pack_size = 32 // weight_bits
    num_packs = _ceil_divide(num_groups, pack_size)
    if group_size >= 128:
        factor = 1
    elif group_size == 64:
        factor = 2
    elif group_size == 32:
        factor = 4
    else:
        raise NotImplementedError
    return _ceil_divide(num_packs, factor) * factor * pack_size


def pack_w4(weight: torch.Tensor) -> torch.Tensor:
    """codes [oc, ic] int32 in 0..15 -> [oc / 4, ic] int16 (tinychat/utils.py:45-54), on the GPU."""
    assert weight.dtype == torch.int32, f"quantized weight should be torch.int32, but got {weight.dtype}."
    if not weight.is_cuda:
        raise DcqError("deepcompressor_amd runs on MI355X GPUs only (there is no CPU path)")
    oc, ic = weight.shape
    assert ic % 32 == 0, "input channel size should be divisible by 32."
    assert oc % 4 == 0 and ic % 64 == 0, "TinyChat packs 4 output x 64 input channels at a time."
    codes = weight.to(torch.int8).contiguous()
    out = torch.empty((oc // 4, ic), dtype=torch.int16, device=weight.device)
    check(lib().dcq_pack_tinychat_w4(C.c_void_p(codes.data_ptr()), oc, ic, C.c_void_p(out.data_ptr()),
                                     stream_of(weight.device)), "dcq_pack_tinychat_w4")
    return out


def convert_to_tinychat_w4x16y16_linear_weight(weight: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor,
                                               zero_pre_scaled: bool = False):
    """(packed weight, scales [groups', oc], negated scaled zeros [groups', oc]) -- tinychat/utils.py:57-107."""
    dtype, device = weight.dtype, weight.device
    assert dtype in (torch.float16, torch.bfloat16), "currently tinychat only supports fp16 and bf16."
    assert scale is not None, "scale tensor is required for quantization."
    assert zero is not None, "zero point tensor is required for quantization."
    weight = weight.to(dtype=torch.float32)
    scale = scale.to(dtype=torch.float32, device=device)
    zero = zero.to(dtype=torch.float32, device=device)
    if zero_pre_scaled:
        zero = zero * scale
    oc, ic = weight.shape
    if scale.numel() == 1:
        scale = scale.view(1, 1).expand(oc, 1)
        ng, gs = 1, ic
    else:
        ng = scale.numel() // oc
        gs = ic // ng
    scale = scale.reshape(oc, ng).contiguous().view(oc, ng, 1)
    assert ic == gs * ng, "input channel size should be equal to group size times number of groups."
    if zero.numel() == 1:
        zero = zero.view(1, 1).expand(oc, ng)
    zero = zero.reshape(oc, ng).contiguous().view(oc, ng, 1)
    weight = weight.view(oc, ng, -1).add_(zero).div_(scale).round_().view(oc, ic)
    assert weight.min() >= 0 and weight.max() <= 15, "quantized weight should be in [0, 15]."
    packed = pack_w4(weight.to(torch.int32))
    rows = ceil_num_groups(ic, gs, weight_bits=4)
    out_scale = torch.zeros((rows, oc), dtype=dtype, device=device)
    out_zero = torch.zeros((rows, oc), dtype=dtype, device=device)
    out_scale[:ng] = scale.view(oc, ng).t().to(dtype=dtype)
    out_zero[:ng] = zero.view(oc, ng).t().to(dtype=dtype).neg_()
    return packed, out_scale, out_zero


def convert_to_tinychat_w4x16y16_linear_state_dict(param_name: str, weight: torch.Tensor, scale: torch.Tensor,
                                                   zero: torch.Tensor, zero_pre_scaled: bool = False
                                                   ) -> dict[str, torch.Tensor]:
    """tinychat/convert.py:14-47."""
    module_name = param_name[:-7]
    weight, scale, zero = convert_to_tinychat_w4x16y16_linear_weight(weight, scale=scale, zero=zero,
                                                                    zero_pre_scaled=zero_pre_scaled)
    return {f"{module_name}.qweight": weight.cpu(), f"{module_name}.scales": scale.cpu(),
            f"{module_name}.scaled_zeros": zero.cpu()}


def convert_to_tinychat_state_dict(state_dict: dict[str, torch.Tensor], scale_dict: dict[str, torch.Tensor],
                                   device: torch.device | str | None = None) -> dict[str, torch.Tensor]:
    """``model.pt`` + ``scale.pt`` -> TinyChat tensors (tinychat/convert.py:50-101).  Weights are converted on
    ``device`` (default: the current GPU); as in the reference a weight whose zero points are all 0 has no zero tensor
    to convert with and fails the converter's assertion."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    scales: dict[str, dict[tuple[int, ...], torch.Tensor]] = {}
    zeros: dict[str, tuple[torch.Tensor | None, bool]] = {}
    for name, tensor in scale_dict.items():
        if name.endswith("zero"):
            zero = None if tensor is None or not bool(tensor.ne(0).any()) else tensor
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
        if param_name in scales:
            zero, pre = zeros.get(param_name, (None, False))
            levels = sorted(scales[param_name].items(), key=lambda kv: kv[0])
            assert len(levels) == 1, "more than one scale levels are not supported."
            converted.update(convert_to_tinychat_w4x16y16_linear_state_dict(
                param_name, param.data.to(device), scale=levels[0][1].to(device),
                zero=zero.to(device) if zero is not None else None, zero_pre_scaled=pre))
        elif isinstance(param, torch.Tensor):
            converted[param_name] = param.clone().cpu()
        else:
            converted[param_name] = param
    return converted
