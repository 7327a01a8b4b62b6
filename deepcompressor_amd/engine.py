"""Thin host layer over the C ABI: canonical group layout, descriptor building, kernel calls.

Nothing in here computes on tensors with torch; torch allocates outputs, permutes layouts that the
kernels cannot address directly, and provides the stream.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import Rounding, Step, check, lib, ptr, storage_of, stream_of
from .data.dtype import QuantDataType
from .data.range import LogQuantRange, QuantRange
from .data.zero import ZeroPointDomain

__all__ = ["canonical_groups", "group_range", "make_rounding", "make_step", "fake_quant_fused", "range_to_scale",
           "scale_level", "scale_join", "zero_point", "quantize_groups", "dequantize_groups", "simple_round",
           "pow_table", "range_search_groups", "gather_columns"]

_ZD = {None: _lib.ZERO_NONE, ZeroPointDomain.PreScale: _lib.ZERO_PRE, ZeroPointDomain.PostScale: _lib.ZERO_POST}
_TORCH_ROUND = {torch.float32: _lib.ROUND_F32, torch.float16: _lib.ROUND_F16, torch.bfloat16: _lib.ROUND_BF16}


# When the quantizer refuses NaN / Inf (the reference asserts on the codes of every call,
# quantizer/impl/base.py:250-251).  The kernels notice a NaN / Inf while they read the tensor anyway and set a device
# flag (no extra pass); the policy says when the host looks at it:
#   "always"   -- before ``quantize`` returns, as the reference (one blocking 4-byte read per call);
#   "deferred" -- never by itself: the flag accumulates until ``raise_if_nonfinite()`` (pipelines that must not block
#                 per call check once per layer or stage);
#   "export"   -- like "always" for calls that return codes, like "deferred" for the others;
#   "never"    -- the kernels are not even given the flag.
FINITE_CHECKS: str = os.environ.get("DCQ_FINITE_CHECKS", "always")
NONFINITE_SEEN = 4


class _FiniteState(threading.local):
    """Per host thread (pipelines on several HIP streams run one thread each): the flag word per device, and the flag
    the kernels launched by this thread currently report into (None outside ``finite_scope``)."""

    def __init__(self):
        self.flags: dict[torch.device, torch.Tensor] = {}
        self.armed: torch.Tensor | None = None


_finite = _FiniteState()


def _flag_of(device: torch.device) -> torch.Tensor:
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    flag = _finite.flags.get(device)
    if flag is None:
        flag = _finite.flags[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return flag


class finite_scope:
    """Kernels launched by this thread inside the scope report NaN / Inf inputs into the thread's flag word of
    ``device`` (policy "never": no flag, nothing reported)."""

    def __init__(self, device: torch.device):
        self.flag = None if FINITE_CHECKS == "never" else _flag_of(device)

    def __enter__(self):
        self.previous, _finite.armed = _finite.armed, self.flag
        return self

    def __exit__(self, *exc):
        _finite.armed = self.previous
        return False


def _nonfinite_ptr():
    flag = _finite.armed
    return None if flag is None else C.c_void_p(flag.data_ptr())


def take_nonfinite(device: torch.device) -> int:
    """Blocking read of the thread's flag word on ``device``; a set flag is cleared.  0: every tensor the armed
    kernels have read since the last look was finite."""
    flag = _finite.flags.get(torch.device(device))
    if flag is None:
        return 0
    seen = int(flag.item())
    if seen:
        flag.zero_()
    return seen


def raise_if_nonfinite(device: torch.device | None = None) -> None:
    """The deferred form of the reference's assertion: raises when a kernel armed by this thread has read a NaN or an
    Inf since the last look (on ``device``, default: every device this thread has used)."""
    devices = list(_finite.flags) if device is None else [torch.device(device)]
    for dev in devices:
        if take_nonfinite(dev):
            raise AssertionError("Quantized tensor contains NaN or Inf (seen by the quantization kernels).")


def _require_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise _lib.DcqError(f"deepcompressor_amd runs on MI355X GPUs only; got a tensor on '{t.device}' "
                            "(there is no CPU path)")


# ---------------------------------------------------------------------------------------------
# layout
# ---------------------------------------------------------------------------------------------
def groups_are_contiguous(view_shape) -> bool:
    """True when every group of a (#g0, gs0, #g1, gs1, ...) view is one contiguous run of a contiguous tensor."""
    counts, sizes = view_shape[0::2], view_shape[1::2]
    split = next((i for i, gs in enumerate(sizes) if gs > 1), len(sizes))
    return all(c == 1 for c in counts[split + 1 :])


def canonical_groups(tensor: torch.Tensor, view_shape):
    """tensor -> (x[G][S] contiguous, restore) where restore maps an [G][S]-shaped result back to tensor.shape."""
    view_shape = tuple(view_shape)
    counts, sizes = view_shape[0::2], view_shape[1::2]
    G, S = 1, 1
    for c in counts:
        G *= c
    for s in sizes:
        S *= s
    shape = tensor.shape
    if groups_are_contiguous(view_shape):
        return tensor.contiguous().view(G, S), (lambda y: y.view(shape))
    n = len(view_shape)
    perm = list(range(0, n, 2)) + list(range(1, n, 2))
    inverse = [perm.index(i) for i in range(n)]
    x = tensor.contiguous().view(view_shape).permute(perm).reshape(G, S)
    return x, (lambda y: y.view(*counts, *sizes).permute(inverse).reshape(shape))


# ---------------------------------------------------------------------------------------------
# descriptors
# ---------------------------------------------------------------------------------------------
_codebooks: dict[tuple[str, torch.device], torch.Tensor] = {}


def _device_codebook(dtype: QuantDataType, device: torch.device) -> torch.Tensor:
    key = (dtype.name, device)
    if key not in _codebooks:
        _codebooks[key] = dtype.get_codebook().values.to(device=device, dtype=torch.float32).contiguous()
    return _codebooks[key]


def make_rounding(dtype, *, has_zero_point: bool, quant_range: QuantRange | None, device: torch.device) -> Rounding:
    """Descriptor of one ``simple_quantize(…, quant_dtype=dtype)`` call."""
    r = Rounding()
    extra = quant_range is not None and quant_range.is_set()
    if isinstance(dtype, torch.dtype):
        if dtype not in _TORCH_ROUND:
            raise NotImplementedError(f"scale dtype {dtype} is not supported (float32, float16, bfloat16 are)")
        r.kind = _TORCH_ROUND[dtype]
    elif dtype.is_exponent:
        log_range = LogQuantRange.construct(dtype, quant_range)
        r.kind, r.lo, r.hi = _lib.ROUND_EXP, log_range.min, log_range.max
        extra = False
    elif dtype.is_float_point:
        table = _device_codebook(dtype, device)
        r.kind, r.lo, r.hi = _lib.ROUND_CODEBOOK, dtype.min_value, dtype.max_value
        r.codebook, r.codebook_size = table.data_ptr(), table.numel()
    else:
        q = QuantRange.construct(dtype, has_zero_point=has_zero_point, quant_range=quant_range)
        r.kind, r.lo, r.hi = _lib.ROUND_INT, q.min, q.max
        extra = False
    if extra:
        lo = quant_range.min if quant_range.min is not None else float("-inf")
        hi = quant_range.max if quant_range.max is not None else float("inf")
        r.has_clamp, r.clamp_lo, r.clamp_hi = 1, lo, hi
    return r


def make_step(*, quant_dtype: QuantDataType, zero_domain, tensor_quant_range: QuantRange, quant_range, scale_dtype,
              zero_dtype, range_bound, q_span: float, device: torch.device) -> Step:
    st = Step()
    zp = zero_domain is not None
    st.tensor = make_rounding(quant_dtype, has_zero_point=zp, quant_range=quant_range, device=device)
    st.scale = make_rounding(scale_dtype, has_zero_point=False, quant_range=None, device=device)
    if zp:
        st.zero = make_rounding(zero_dtype, has_zero_point=True, quant_range=None, device=device)
    st.zero_domain = _ZD[zero_domain]
    st.q_span = q_span
    st.q_min = tensor_quant_range.min
    if range_bound is not None:
        if range_bound.min is not None:
            st.has_bound_min, st.bound_min = 1, range_bound.min
        if range_bound.max is not None:
            st.has_bound_max, st.bound_max = 1, range_bound.max
    return st


def _f32(n, device) -> torch.Tensor:
    return torch.empty(n, dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------------------------
# kernel calls
# ---------------------------------------------------------------------------------------------
def fake_quant_fused(step: Step, x: torch.Tensor, *, range_max=None, range_min=None, ratio=None, want_dequant=True,
                     want_codes=False, want_scales=True):
    """x[G][S] -> (dequant | None, codes | None, scale[G], zero[G]).  ``want_scales`` False: the per-group scales and
    zeros are not written (and returned as None): a caller that only takes the fake-quantized tensor saves their
    buffers and, for short groups, a noticeable share of the pass's writes (8 bytes per group)."""
    _require_gpu(x)
    G, S = x.shape
    dq = torch.empty_like(x) if want_dequant else None
    codes = torch.empty((G, S), dtype=torch.float32, device=x.device) if want_codes else None
    scale, zero = (_f32(G, x.device), _f32(G, x.device)) if want_scales else (None, None)
    check(lib().dcq_fake_quant_groups(C.byref(step), ptr(x), storage_of(x.dtype), G, S, ptr(range_max), ptr(range_min),
                                      int(ratio is not None), float(ratio or 0.0), ptr(dq), ptr(codes), ptr(scale),
                                      ptr(zero), _nonfinite_ptr(), stream_of(x.device)), "dcq_fake_quant_groups")
    return dq, codes, scale, zero


def fake_quant_fused_outer(step: Step, x: torch.Tensor, outer: torch.Tensor, outer_lo: float, outer_hi: float, *,
                           deq_clamp: bool, deq_lo: float, deq_hi: float, range_max=None, range_min=None, ratio=None,
                           out_dtype=None, want_dequant=True, want_codes=False, want_scales=True):
    """Last step of a progressive quantizer in one pass: x[G][S], outer[G] (the previous steps' scale per group)
    -> (dequant | None, codes | None, scale[G] | None, zero[G] | None)."""
    _require_gpu(x)
    G, S = x.shape
    out_dtype = out_dtype or x.dtype
    dq = torch.empty((G, S), dtype=out_dtype, device=x.device) if want_dequant else None
    codes = torch.empty((G, S), dtype=torch.float32, device=x.device) if want_codes else None
    scale, zero = (_f32(G, x.device), _f32(G, x.device)) if want_scales else (None, None)
    check(lib().dcq_fake_quant_groups_outer(C.byref(step), ptr(x), storage_of(x.dtype), G, S, ptr(outer), float(outer_lo),
                                            float(outer_hi), int(deq_clamp), float(deq_lo), float(deq_hi),
                                            ptr(range_max), ptr(range_min), int(ratio is not None), float(ratio or 0.0),
                                            ptr(dq), storage_of(out_dtype), ptr(codes), ptr(scale), ptr(zero),
                                            _nonfinite_ptr(), stream_of(x.device)), "dcq_fake_quant_groups_outer")
    return dq, codes, scale, zero


def progressive_is_fusable(first: Step, last: Step, row_len: int, group: int, dtype: torch.dtype) -> bool:
    """Whether ``dcq_fake_quant_progressive`` covers a two-step quantizer on rows of ``row_len`` values with groups of
    ``group`` in the last step (the limits documented at its declaration)."""
    vec = 4 if dtype == torch.float32 else 8
    lanes = group // vec
    if group % vec or row_len % group or lanes > 64 or lanes & (lanes - 1) or row_len // vec > 2048:
        return False
    if first.zero_domain != _lib.ZERO_NONE or first.scale.kind == _lib.ROUND_CODEBOOK:
        return False
    kinds = (_lib.ROUND_F32, _lib.ROUND_F16, _lib.ROUND_BF16, _lib.ROUND_INT)
    if last.tensor.kind != _lib.ROUND_INT or last.scale.kind not in kinds or last.scale.has_clamp:
        return False
    if last.zero_domain == _lib.ZERO_PRE:
        return last.zero.kind == _lib.ROUND_INT
    if last.zero_domain == _lib.ZERO_POST:
        return last.zero.kind == last.scale.kind and not last.zero.has_clamp
    return True


def fake_quant_progressive(first: Step, last: Step, x: torch.Tensor, group: int, outer_lo: float, outer_hi: float, *,
                           deq_clamp: bool, deq_lo: float, deq_hi: float, first_ratio=None, ratio=None,
                           want_dequant=True, want_codes=False, want_scales=True):
    """Both steps of a two-step quantizer in one pass: x[rows][row_len] -> (dequant | None, codes | None,
    first_scale[rows], scale[G] | None, zero[G] | None) with G = rows * row_len / group."""
    _require_gpu(x)
    rows, row_len = x.shape
    G = rows * (row_len // group)
    dq = torch.empty_like(x) if want_dequant else None
    codes = torch.empty((rows, row_len), dtype=torch.float32, device=x.device) if want_codes else None
    s0 = _f32(rows, x.device)
    scale, zero = (_f32(G, x.device), _f32(G, x.device)) if want_scales else (None, None)
    check(lib().dcq_fake_quant_progressive(
        C.byref(first), C.byref(last), ptr(x), storage_of(x.dtype), rows, row_len, group, int(first_ratio is not None),
        float(first_ratio or 0.0), int(ratio is not None), float(ratio or 0.0), float(outer_lo), float(outer_hi),
        int(deq_clamp), float(deq_lo), float(deq_hi), ptr(dq), ptr(codes), ptr(s0), ptr(scale), ptr(zero),
        _nonfinite_ptr(), stream_of(x.device)), "dcq_fake_quant_progressive")
    return dq, codes, s0, scale, zero


def fake_quant_two_level(step: Step, level0: Rounding, level0_span: float, x: torch.Tensor, *, ratio=None,
                         want_dequant=True, want_codes=False, want_scales=True):
    """Two scale levels, the first per tensor: x[G][S] -> (dequant | None, codes | None, level0[1], level1[G] | None,
    compound scale[G] | None)."""
    _require_gpu(x)
    G, S = x.shape
    dq = torch.empty_like(x) if want_dequant else None
    codes = torch.empty((G, S), dtype=torch.float32, device=x.device) if want_codes else None
    top, first = _f32(1, x.device), _f32(1, x.device)
    second, scale = (_f32(G, x.device), _f32(G, x.device)) if want_scales else (None, None)
    check(lib().dcq_fake_quant_groups_two_level(C.byref(step), C.byref(level0), float(level0_span), ptr(x),
                                                storage_of(x.dtype), G, S, int(ratio is not None), float(ratio or 0.0),
                                                ptr(top), ptr(dq), ptr(codes), ptr(first), ptr(second), ptr(scale),
                                                _nonfinite_ptr(), stream_of(x.device)),
          "dcq_fake_quant_groups_two_level")
    return dq, codes, first, second, scale


def group_range(x: torch.Tensor, *, with_min: bool, outer: torch.Tensor | None = None, outer_lo: float = 0.0,
                outer_hi: float = 0.0):
    _require_gpu(x)
    G, S = x.shape
    vmax = _f32(G, x.device)
    vmin = _f32(G, x.device) if with_min else None
    zd = _lib.ZERO_POST if with_min else _lib.ZERO_NONE
    check(lib().dcq_group_range(ptr(x), storage_of(x.dtype), G, S, zd, ptr(outer), outer_lo, outer_hi, ptr(vmax),
                                ptr(vmin), _nonfinite_ptr(), stream_of(x.device)), "dcq_group_range")
    return vmax, vmin


def range_to_scale(step: Step, vmax, vmin, *, ratio=None, static_max=None, static_min=None, vavg=None):
    """-> (linear_scale[n], vmin'[n] | None).  vmin' is the minimum the zero point must use.  ``vavg``: the group
    means, for a floating-point type with a zero point (range symmetric about the mean)."""
    ref = vmax if vmax is not None else static_max
    n = ref.numel()
    lin = _f32(n, ref.device)
    zp = step.zero_domain != _lib.ZERO_NONE
    vmin_io = None
    if zp:
        vmin_io = vmin.clone() if vmin is not None else _f32(n, ref.device)
    check(lib().dcq_range_to_scale(C.byref(step), n, ptr(vmax), ptr(vmin_io), ptr(vavg), int(ratio is not None),
                                   float(ratio or 0.0), ptr(static_max), ptr(static_min), ptr(lin),
                                   stream_of(ref.device)), "dcq_range_to_scale")
    return lin, vmin_io


def scale_level(rounding: Rounding, s: torch.Tensor, P: int, Q: int, *, reduce: bool, span: float) -> torch.Tensor:
    """One scale level on s[P][Q] (modified in place when reducing); returns the level tensor [P]."""
    level = _f32(P, s.device)
    check(lib().dcq_scale_level(C.byref(rounding), ptr(s), P, Q, int(reduce), float(span), ptr(level),
                                stream_of(s.device)), "dcq_scale_level")
    return level


def simple_round(rounding: Rounding, values: torch.Tensor) -> torch.Tensor:
    """rounding(|values|) element-wise on a float32 tensor (scales are non-negative by construction)."""
    flat = values.contiguous().view(-1)
    return scale_level(rounding, flat, flat.numel(), 1, reduce=False, span=1.0).view(values.shape)


def simple_round_signed(rounding: Rounding, values: torch.Tensor) -> torch.Tensor:
    """rounding(values) element-wise on a float32 tensor of any sign (int / codebook / torch-dtype kinds)."""
    _require_gpu(values)
    flat = values.contiguous().view(-1)
    out = torch.empty_like(flat)
    check(lib().dcq_round_values(C.byref(rounding), ptr(flat), ptr(out), flat.numel(), stream_of(flat.device)),
          "dcq_round_values")
    return out.view(values.shape)


def assert_finite(values: torch.Tensor, what: str, *, inf_first: bool = False) -> None:
    """``assert not isnan().any()`` + ``assert not isinf().any()`` of the reference (same messages) as one pass and one
    host read when everything is finite, which is the case that matters for speed."""
    if values.is_cuda and values.dtype == torch.float32 and values.is_contiguous() and values.data_ptr() % 16 == 0:
        flags = torch.zeros(1, dtype=torch.int32, device=values.device)
        check(lib().dcq_nonfinite_flags(ptr(values), values.numel(), ptr(flags), stream_of(values.device)),
              "dcq_nonfinite_flags")
        seen = int(flags.item())
        if seen == 0:
            return
        order = (("Inf", 2), ("NaN", 1)) if inf_first else (("NaN", 1), ("Inf", 2))
        for name, bit in order:
            assert not seen & bit, f"{what} contains {name}."
    if bool(torch.isfinite(values).all()):
        return
    checks = [(values.isnan, "NaN"), (values.isinf, "Inf")]
    for probe, name in (reversed(checks) if inf_first else checks):
        assert not probe().any(), f"{what} contains {name}."


def gather_columns(src: torch.Tensor, index: torch.Tensor, zero_columns: torch.Tensor | None = None) -> torch.Tensor:
    """``torch.where(zero_columns, 0, src.float())[:, index]`` for a float32 / float16 / bfloat16 matrix
    (``zero_columns``: bool per source column) as a float32 matrix, one pass with each row staged in LDS; shapes the
    kernel does not take (rows that are not a multiple of 4 long or longer than 36864) go through the torch
    expression."""
    _require_gpu(src)
    assert src.ndim == 2 and src.dtype in _TORCH_ROUND and index.dtype == torch.int64 and index.numel() == src.shape[1]
    R, Cn = src.shape
    src = src.contiguous()
    if Cn % 4 or Cn > 36864 or src.data_ptr() % 16:
        src = src.to(torch.float32)
        if zero_columns is not None:
            src = torch.where(zero_columns, torch.zeros((), device=src.device, dtype=src.dtype), src)
        return src[:, index].contiguous()
    index = index.contiguous()
    mask = None if zero_columns is None else zero_columns.contiguous().view(torch.uint8)
    out = torch.empty((R, Cn), dtype=torch.float32, device=src.device)
    check(lib().dcq_gather_columns(ptr(src), storage_of(src.dtype), Cn, R, Cn, ptr(index), ptr(mask), ptr(out), Cn,
                                   stream_of(src.device)), "dcq_gather_columns")
    return out


def scale_join(a: torch.Tensor, b: torch.Tensor | None, *, fix_zero: bool) -> torch.Tensor:
    out = torch.empty_like(a)
    check(lib().dcq_scale_join(ptr(a), ptr(b), a.numel(), 1, int(fix_zero), ptr(out), stream_of(a.device)),
          "dcq_scale_join")
    return out


def zero_point(step: Step, vmin: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    zero = torch.empty_like(scale)
    check(lib().dcq_zero_point(C.byref(step), scale.numel(), ptr(vmin), ptr(scale), ptr(zero),
                               stream_of(scale.device)), "dcq_zero_point")
    return zero


def quantize_groups(step: Step, x: torch.Tensor, scale, zero, *, outer=None, outer_lo=0.0, outer_hi=0.0,
                    deq_clamp=False, deq_lo=0.0, deq_hi=0.0, out_dtype=None, want_dequant=True, want_codes=False,
                    round_delta=None):
    """``round_delta``: float32 [G][S] (canonical group order), codes = clamp(floor(t) + delta)."""
    _require_gpu(x)
    G, S = x.shape
    if round_delta is not None:
        assert round_delta.shape == (G, S) and round_delta.dtype == torch.float32 and round_delta.is_contiguous()
    out_dtype = out_dtype or x.dtype
    dq = torch.empty((G, S), dtype=out_dtype, device=x.device) if want_dequant else None
    codes = torch.empty((G, S), dtype=torch.float32, device=x.device) if want_codes else None
    check(lib().dcq_quantize_groups(C.byref(step), ptr(x), storage_of(x.dtype), G, S, ptr(scale), ptr(zero), ptr(outer),
                                    outer_lo, outer_hi, int(deq_clamp), deq_lo, deq_hi, ptr(dq),
                                    storage_of(out_dtype), ptr(codes), ptr(round_delta), _nonfinite_ptr(),
                                    stream_of(x.device)), "dcq_quantize_groups")
    return dq, codes


def prescale_groups(x: torch.Tensor, outer: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """clamp(x[g] / outer[g], lo, hi) as float32 [G][S]."""
    _require_gpu(x)
    G, S = x.shape
    assert outer.numel() == G and outer.dtype == torch.float32 and outer.is_contiguous(), "one divisor per group"
    out = torch.empty((G, S), dtype=torch.float32, device=x.device)
    check(lib().dcq_prescale_groups(ptr(x), storage_of(x.dtype), G, S, ptr(outer), lo, hi, ptr(out),
                                    stream_of(x.device)), "dcq_prescale_groups")
    return out


def dequantize_groups(step: Step, codes: torch.Tensor, scale, zero, *, outer=None, deq_clamp=False, deq_lo=0.0,
                      deq_hi=0.0, out_dtype=torch.float32) -> torch.Tensor:
    _require_gpu(codes)
    G, S = codes.shape
    dq = torch.empty((G, S), dtype=out_dtype, device=codes.device)
    check(lib().dcq_dequantize_groups(C.byref(step), ptr(codes), G, S, ptr(scale), ptr(zero), ptr(outer),
                                      int(deq_clamp), deq_lo, deq_hi, ptr(dq), storage_of(out_dtype),
                                      stream_of(codes.device)), "dcq_dequantize_groups")
    return dq


_pow_tables: dict[tuple[torch.device, torch.dtype, float], tuple[torch.Tensor, torch.cuda.Event]] = {}
_pow_tables_lock = threading.Lock()


def pow_table(device: torch.device, dtype: torch.dtype, degree: float) -> torch.Tensor:
    """The 32768-entry table of ``x.pow_(degree)`` over all non-negative bit patterns of a 16-bit dtype.

    The table is filled by a kernel on the stream of whichever host thread asks first; the cache is shared by all
    threads (one HIP stream each under ``parallel.run_on_streams``), so every caller's current stream is made to
    wait for the event recorded behind that kernel before it may launch a consumer."""
    key = (device, dtype, float(degree))
    with _pow_tables_lock:
        entry = _pow_tables.get(key)
        if entry is None:
            table = torch.empty(32768, dtype=torch.int16, device=device)
            stream = torch.cuda.current_stream(device)
            check(lib().dcq_pow_table(storage_of(dtype), float(degree), ptr(table), stream_of(device)),
                  "dcq_pow_table")
            ready = torch.cuda.Event()
            ready.record(stream)
            entry = _pow_tables[key] = (table, ready)
    table, ready = entry
    if not ready.query():
        torch.cuda.current_stream(device).wait_event(ready)
    return table


_ratio_lists: dict[tuple, torch.Tensor] = {}


def _device_ratios(ratios: tuple, device: torch.device) -> torch.Tensor:
    """The candidate ratios as a float32 device vector, kept per (list, device): every linear of a stage searches the
    same list, and a copy from pageable host memory waits for the work already queued on the stream."""
    key = (ratios, device)
    r = _ratio_lists.get(key)
    if r is None:
        if len(_ratio_lists) > 64:
            _ratio_lists.clear()
        r = _ratio_lists[key] = torch.tensor(ratios, dtype=torch.float64).to(torch.float32).to(device)
    return r


def range_search_groups(step: Step, w: torch.Tensor, ratios: list[float], degree: float, *, want_errors=False,
                        unrounded_errors=False):
    """w[G][S] -> dict(max, min | None, ratio, error[, errors[K][G]]) of float32 tensors.  ``errors`` holds the
    per-candidate group errors rounded to the weight dtype, or the float32 sums before that rounding."""
    _require_gpu(w)
    G, S = w.shape
    dev = w.device
    r = _device_ratios(tuple(ratios), dev)
    zp = step.zero_domain != _lib.ZERO_NONE
    out = dict(max=_f32(G, dev), min=_f32(G, dev) if zp else None, ratio=_f32(G, dev), error=_f32(G, dev))
    errors = torch.empty((len(ratios), G), dtype=torch.float32, device=dev) if want_errors else None
    table = pow_table(dev, w.dtype, degree) if w.dtype != torch.float32 else None
    check(lib().dcq_range_search_groups(C.byref(step), ptr(w), storage_of(w.dtype), G, S, ptr(r), len(ratios),
                                        float(degree), ptr(table), ptr(out["max"]), ptr(out["min"]),
                                        ptr(out["ratio"]), ptr(out["error"]), ptr(errors),
                                        1 if unrounded_errors else 0, stream_of(dev)),
          "dcq_range_search_groups")
    if want_errors:
        out["errors"] = errors
    return out


def grouped_products(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """a [M][B * K] (row-major), b [B][K][N] -> out [B][M][N] with out[g] = a[:, g * K : (g + 1) * K] @ b[g]: the
    partial products of every group of K columns (B = 1: a plain product).  Accumulated in float64 on the matrix
    cores and rounded double -> float32 -> dtype once (``dcq_gemm_f64``), so each product is the correctly rounded
    float32 sum, cast like ``Tensor.to`` -- whatever order a float32-accumulating GEMM would have used, and
    reproducible on any machine."""
    _require_gpu(a)
    assert a.ndim == 2 and b.ndim == 3 and a.stride(1) == 1 and b.is_contiguous() and a.device == b.device
    groups, K, N = b.shape
    M = a.shape[0]
    assert a.shape[1] == groups * K, f"{tuple(a.shape)} does not split into {groups} groups of {K} columns"
    out_dtype = out_dtype or a.dtype
    out = torch.empty((groups, M, N), dtype=out_dtype, device=a.device)
    if out.numel() == 0:
        return out
    storage = {torch.float64: 3, torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    raw = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    check(lib().dcq_gemm_f64(raw(a), storage[a.dtype], a.stride(0), 0, K, raw(b), storage[b.dtype], N, K * N, raw(out),
                             storage[out_dtype], N, M * N, M, N, K, groups, stream_of(a.device)), "dcq_gemm_f64")
    return out


def smooth_step_is_fusable(step: Step) -> bool:
    """Whether ``dcq_smooth_fake_quant_groups`` covers this step: an integer type with one float scale level."""
    floats = (_lib.ROUND_F32, _lib.ROUND_F16, _lib.ROUND_BF16)
    if step.tensor.kind != _lib.ROUND_INT or step.scale.kind not in floats or step.scale.has_clamp:
        return False
    if step.zero_domain == _lib.ZERO_PRE:
        return step.zero.kind == _lib.ROUND_INT
    if step.zero_domain == _lib.ZERO_POST:
        return step.zero.kind == step.scale.kind and not step.zero.has_clamp
    return True


def smooth_fake_quant(step: Step, x: torch.Tensor, groups_per_row: int, chan_scale: torch.Tensor, *,
                      divide_first: bool, ratio: float | None = None, errors_degree: float | None = None
                      ) -> torch.Tensor:
    """x[G][S] (groups along the channels) and chan_scale [C] or [K][C] float32 -> the smoothed, fake-quantized,
    un-smoothed tensor in x's dtype (one candidate), or with ``errors_degree`` the float32 [K][G] sums of
    |processed - x|^degree of every candidate, evaluated in x's dtype."""
    _require_gpu(x)
    assert x.ndim == 2 and x.is_contiguous()
    G, S = x.shape
    cs = chan_scale.to(device=x.device, dtype=torch.float32).contiguous().view(-1, groups_per_row * S)
    K = cs.shape[0]
    out = errors = table = None
    degree = 0.0
    if errors_degree is None:
        assert K == 1, "one candidate when the processed tensor is wanted"
        out = torch.empty_like(x)
    else:
        degree = float(errors_degree)
        errors = torch.empty((K, G), dtype=torch.float32, device=x.device)
        table = pow_table(x.device, x.dtype, degree) if x.dtype != torch.float32 else None
    check(lib().dcq_smooth_fake_quant_groups(C.byref(step), ptr(x), storage_of(x.dtype), G, S, groups_per_row, ptr(cs), K,
                                             1 if divide_first else 0, 0 if ratio is None else 1,
                                             0.0 if ratio is None else float(ratio), ptr(out), ptr(errors), degree,
                                             ptr(table), _nonfinite_ptr(), stream_of(x.device)),
          "dcq_smooth_fake_quant_groups")
    return out if errors is None else errors


def smooth_codes_are_fusable(step: Step) -> bool:
    """Whether ``dcq_smooth_quantize_groups`` covers this step: fusable, and the integer its scale multiplies (the
    code, minus a zero point of the code's domain) fits a signed byte."""
    if not smooth_step_is_fusable(step) or step.zero_domain == _lib.ZERO_POST:
        return False
    pre = step.zero_domain == _lib.ZERO_PRE
    top = step.tensor.hi - (step.zero.lo if pre else 0.0)
    bottom = step.tensor.lo - (step.zero.hi if pre else 0.0)
    return top <= 127.0 and bottom >= -128.0


def smooth_quantize(step: Step, x: torch.Tensor, groups_per_row: int, chan_scale: torch.Tensor, *,
                    divide_first: bool, ratio: float | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """x[G][S] and chan_scale [K][C] float32 -> (int8 codes [K][G][S], float32 scales [K][G]) of every smoothing
    candidate: the integers and group scales whose product, smoothed back, is what ``smooth_fake_quant`` writes."""
    _require_gpu(x)
    assert x.ndim == 2 and x.is_contiguous()
    G, S = x.shape
    cs = chan_scale.to(device=x.device, dtype=torch.float32).contiguous().view(-1, groups_per_row * S)
    K = cs.shape[0]
    codes = torch.empty((K, G, S), dtype=torch.int8, device=x.device)
    scales = torch.empty((K, G), dtype=torch.float32, device=x.device)
    check(lib().dcq_smooth_quantize_groups(C.byref(step), ptr(x), storage_of(x.dtype), G, S, groups_per_row, ptr(cs), K,
                                           1 if divide_first else 0, 0 if ratio is None else 1,
                                           0.0 if ratio is None else float(ratio), ptr(codes), ptr(scales),
                                           _nonfinite_ptr(), stream_of(x.device)), "dcq_smooth_quantize_groups")
    return codes, scales


def linear_candidate_errors(x_codes: torch.Tensor, x_scales: torch.Tensor, w_codes: torch.Tensor,
                            w_scales: torch.Tensor, bias: torch.Tensor | None, target: torch.Tensor,
                            out: torch.Tensor | None = None) -> torch.Tensor:
    """Codes [K][T][C] / [K][R][C] (int8) with their group scales [K][T][gx] / [K][R][gw] and the outputs to match
    [T][R] -> float64 [K]: sum over the outputs of (x' w'^T + bias - target)^2 per candidate, added to ``out`` when
    given (``dcq_linear_candidate_errors``)."""
    _require_gpu(x_codes)
    K, T, Cn = x_codes.shape
    R = w_codes.shape[1]
    assert w_codes.shape == (K, R, Cn) and x_codes.dtype == torch.int8 and w_codes.dtype == torch.int8
    assert x_codes.is_contiguous() and w_codes.is_contiguous() and x_scales.is_contiguous() and w_scales.is_contiguous()
    assert x_scales.dtype == torch.float32 and w_scales.dtype == torch.float32
    gx, gw = x_scales.numel() // (K * T), w_scales.numel() // (K * R)
    assert x_scales.numel() == K * T * gx and w_scales.numel() == K * R * gw and Cn % gx == 0 and Cn % gw == 0
    assert target.is_cuda and target.dtype == torch.float32 and target.shape == (T, R) and target.stride(1) == 1
    if bias is not None:
        bias = bias.to(device=x_codes.device, dtype=torch.float32).contiguous()
        assert bias.numel() == R
    accumulate = out is not None
    if out is None:
        out = torch.empty(K, dtype=torch.float64, device=x_codes.device)
    assert out.dtype == torch.float64 and out.numel() == K and out.is_contiguous()
    need = lib().dcq_linear_candidate_errors_workspace(T, R, K)
    work = torch.empty(max(need, 8), dtype=torch.uint8, device=x_codes.device)
    check(lib().dcq_linear_candidate_errors(ptr(x_codes), ptr(x_scales), T, Cn // gx, ptr(w_codes), ptr(w_scales), R,
                                            Cn // gw, Cn, K, ptr(bias), C.c_void_p(target.data_ptr()), target.stride(0),
                                            ptr(work), need,
                                            ptr(out), int(accumulate), stream_of(x_codes.device)),
          "dcq_linear_candidate_errors")
    return out


def group_error(a: torch.Tensor, b: torch.Tensor, degree: float) -> torch.Tensor:
    """a, b [G][S] of one dtype -> float32 [G]: sum over each group of |a - b|^degree evaluated in that dtype,
    accumulated in float32 and not rounded back."""
    _require_gpu(a)
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    G, S = a.shape
    out = _f32(G, a.device)
    table = pow_table(a.device, a.dtype, degree) if a.dtype != torch.float32 else None
    check(lib().dcq_group_error(ptr(a), ptr(b), storage_of(a.dtype), G, S, float(degree), ptr(table), ptr(out),
                                stream_of(a.device)), "dcq_group_error")
    return out


def product_error(y: torch.Tensor, y0: torch.Tensor, degree: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """y, y0 [G][S] of one dtype -> float32 [G]: sum_i float32(dtype(y - y0))^degree, added to ``out`` when given."""
    _require_gpu(y)
    assert y.shape == y0.shape and y.dtype == y0.dtype and y.is_contiguous() and y0.is_contiguous()
    G, S = y.shape
    accumulate = out is not None
    if out is None:
        out = _f32(G, y.device)
    assert out.numel() == G and out.dtype == torch.float32 and out.is_contiguous()
    check(lib().dcq_product_error(ptr(y), ptr(y0), storage_of(y.dtype), G, S, float(degree), int(accumulate), ptr(out),
                                  stream_of(y.device)), "dcq_product_error")
    return out
