"""ctypes binding of ``libdcq.so`` (the C ABI declared in ``include/dcq.h``).

There is no fallback: if the shared library is missing, or a tensor is not on a GPU, the calling
operation raises.  torch is used only to own device memory and streams.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

__all__ = ["lib", "check", "Rounding", "Step", "F32", "F16", "BF16", "storage_of", "ptr", "stream_of", "DcqError"]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdcq.so")

F32, F16, BF16 = 0, 1, 2
ZERO_NONE, ZERO_PRE, ZERO_POST = 0, 1, 2
ROUND_F32, ROUND_F16, ROUND_BF16, ROUND_INT, ROUND_CODEBOOK, ROUND_EXP = range(6)


class DcqError(RuntimeError):
    pass


class Rounding(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("lo", C.c_float),
        ("hi", C.c_float),
        ("codebook", C.c_void_p),
        ("codebook_size", C.c_int32),
        ("has_clamp", C.c_int32),
        ("clamp_lo", C.c_float),
        ("clamp_hi", C.c_float),
    ]


class Step(C.Structure):
    _fields_ = [
        ("tensor", Rounding),
        ("scale", Rounding),
        ("zero", Rounding),
        ("zero_domain", C.c_int32),
        ("q_span", C.c_float),
        ("q_min", C.c_float),
        ("has_bound_min", C.c_int32),
        ("has_bound_max", C.c_int32),
        ("bound_min", C.c_float),
        ("bound_max", C.c_float),
    ]


_p, _i32, _i64, _f, _d = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_SIGNATURES = {
    "dcq_round_to_codebook": [_p, _p, _i64, _i32, _p, _i32, _p],
    "dcq_fake_quant_groups": [C.POINTER(Step), _p, _i32, _i64, _i64, _p, _p, _i32, _f, _p, _p, _p, _p, _p, _p],
    "dcq_group_range": [_p, _i32, _i64, _i64, _i32, _p, _f, _f, _p, _p, _p, _p],
    "dcq_range_to_scale": [C.POINTER(Step), _i64, _p, _p, _p, _i32, _f, _p, _p, _p, _p],
    "dcq_scale_level": [C.POINTER(Rounding), _p, _i64, _i64, _i32, _f, _p, _p],
    "dcq_round_values": [C.POINTER(Rounding), _p, _p, _i64, _p],
    "dcq_selftest_division": [_p, _p, _p, _p, _p, _p, _i64, _p],
    "dcq_scale_join": [_p, _p, _i64, _i64, _i32, _p, _p],
    "dcq_zero_point": [C.POINTER(Step), _i64, _p, _p, _p, _p],
    "dcq_quantize_groups": [C.POINTER(Step), _p, _i32, _i64, _i64, _p, _p, _p, _f, _f, _i32, _f, _f, _p, _i32, _p, _p,
                            _p, _p],
    "dcq_prescale_groups": [_p, _i32, _i64, _i64, _p, _f, _f, _p, _p],
    "dcq_dequantize_groups": [C.POINTER(Step), _p, _i64, _i64, _p, _p, _p, _i32, _f, _f, _p, _i32, _p],
    "dcq_pow_table": [_i32, _d, _p, _p],
    "dcq_hessian_accumulate": [_p, _i32, _i64, _i64, _f, _p, _p, _i64, _p],
    "dcq_hessian_accumulate_ex": [_p, _i32, _i64, _i64, _f, _p, _p, _i64, _i32, _p, _p, _p],
    "dcq_gptq_sweep_block": [C.POINTER(Step), _p, _i64, _i64, _i64, _p, _i32, _i32, _p, _p, _f, _i32, _i32, _i32, _p,
                             _p, _p, _i64, _p],
    "dcq_gptq_sweep": [C.POINTER(Step), _p, _i64, _i64, _i64, _p, _p, _p, _f, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _i64,
                       _p],
    "dcq_nonfinite_flags": [_p, _i64, _p, _p],
    "dcq_gather_columns": [_p, _i32, _i64, _i64, _i64, _p, _p, _p, _i64, _p],
    "dcq_pack_qserve_w4": [_p, _i64, _i64, _p, _p],
    "dcq_pack_nunchaku": [_p, _i64, _i64, _i32, _p, _p],
    "dcq_gemm_f32": [_p, _i64, _i32, _p, _i64, _p, _i64, _i64, _i64, _i64, _f, _f, _p],
    "dcq_gemm_f32_two_term": [_p, _i64, _i32, _p, _i64, _p, _i64, _i64, _i64, _i64, _f, _f, _p, _i64, _p],
    "dcq_cholesky_inverse_upper": [_p, _i64, _p, _f, _p, _p, _i64, _p, _p],
    "dcq_cholesky_inverse_upper_static": [_p, _i64, _p, _f, _p, _p, _p],
    "dcq_gram_inverse_factor_f64": [_p, _i64, _i32, _p, _i64, _p, _p],
    "dcq_gemm_f64": [_p, _i32, _i64, _i32, _i64, _p, _i32, _i64, _i64, _p, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "dcq_fake_quant_groups_outer": [C.POINTER(Step), _p, _i32, _i64, _i64, _p, _f, _f, _i32, _f, _f, _p, _p, _i32, _f, _p,
                                    _i32, _p, _p, _p, _p, _p],
    "dcq_fake_quant_progressive": [C.POINTER(Step), C.POINTER(Step), _p, _i32, _i64, _i64, _i64, _i32, _f, _i32, _f, _f, _f,
                                   _i32, _f, _f, _p, _p, _p, _p, _p, _p, _p],
    "dcq_fake_quant_groups_two_level": [C.POINTER(Step), C.POINTER(Rounding), _f, _p, _i32, _i64, _i64, _i32, _f, _p, _p,
                                        _p, _p, _p, _p, _p, _p],
    "dcq_smooth_fake_quant_groups": [C.POINTER(Step), _p, _i32, _i64, _i64, _i64, _p, _i32, _i32, _i32, _f, _p, _p, _d,
                                     _p, _p, _p],
    "dcq_smooth_quantize_groups": [C.POINTER(Step), _p, _i32, _i64, _i64, _i64, _p, _i32, _i32, _i32, _f, _p, _p, _p, _p],
    "dcq_linear_candidate_errors": [_p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _i32, _p, _p, _i64, _p, _i64, _p, _i32,
                                    _p],
    "dcq_channel_metric": [_p, _i32, _i64, _i64, _i64, _i32, _p, _p, _p],
    "dcq_scale_channels": [_p, _i32, _i64, _i64, _i64, _p, _i64, _i32, _p, _i32, _p],
    "dcq_range_search_groups": [C.POINTER(Step), _p, _i32, _i64, _i64, _p, _i32, _d, _p, _p, _p, _p, _p, _p, _i32,
                                _p],
    "dcq_group_error": [_p, _p, _i32, _i64, _i64, _d, _p, _p, _p],
    "dcq_product_error": [_p, _p, _i32, _i64, _i64, _d, _i32, _p, _p],
}
_lib = None


def lib():
    """The loaded library; raises with the build command if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DcqError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or deepcompressor_amd/csrc/build.sh). There is no CPU or torch fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, sig in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes, fn.restype = sig, C.c_int
        handle.dcq_hessian_workspace_bytes.argtypes = [_i64, _i64]
        handle.dcq_hessian_workspace_bytes.restype = C.c_int64
        handle.dcq_cholesky_workspace_bytes.argtypes = [_i64]
        handle.dcq_cholesky_workspace_bytes.restype = C.c_int64
        handle.dcq_linear_candidate_errors_workspace.argtypes = [_i64, _i64, _i32]
        handle.dcq_linear_candidate_errors_workspace.restype = C.c_int64
        handle.dcq_gemm_f32_two_term_workspace_bytes.argtypes = [_i64, _i64, _i64]
        handle.dcq_gemm_f32_two_term_workspace_bytes.restype = C.c_int64
        handle.dcq_gptq_sweep_workspace_bytes.argtypes = [_i64, _i64, _i32]
        handle.dcq_gptq_sweep_workspace_bytes.restype = C.c_int64
        handle.dcq_version.restype = C.c_char_p
        handle.dcq_last_error.restype = C.c_char_p
        _lib = handle
    return _lib


def bind(name: str, argtypes) -> None:
    fn = getattr(lib(), name)
    fn.argtypes, fn.restype = argtypes, C.c_int


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DcqError(f"{what} failed ({rc}): {lib().dcq_last_error().decode()}")


_STORAGE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


def storage_of(dtype: torch.dtype) -> int:
    try:
        return _STORAGE[dtype]
    except KeyError:
        raise TypeError(f"unsupported tensor dtype {dtype}; expected float32, float16 or bfloat16") from None


def ptr(t: torch.Tensor | None):
    """Device address of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DcqError("this engine runs on the GPU only: got a tensor on " + str(t.device))
    if not t.is_contiguous():
        raise DcqError("internal error: non-contiguous buffer handed to the C ABI")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(device: torch.device):
    """The HIP stream torch currently launches on for ``device`` (the raw handle: building a ``torch.cuda.Stream``
    object per call costs several microseconds)."""
    if _raw_stream is not None:
        index = device.index
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if index is None else index))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
