"""GPTQ quantization kernel (reference: deepcompressor/quantizer/kernel/gptq.py:23-285).

Same config fields and kernel interface as the reference.  The Hessian X^T X runs on the matrix cores
(``dcq_hessian_accumulate``), the column sweep of each 128-column block is one HIP kernel with rows as the
parallel axis (``dcq_gptq_sweep``), the inverse Cholesky factor of the damped Hessian is a blocked
factorisation + triangular inverse on the float32 matrix-core GEMM (``dcq_cholesky_inverse_upper``), and the
lazy update between blocks is that same GEMM (``dcq_gemm_f32``).
"""

from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass

import torch

from ... import _lib, engine
from ..._lib import check, lib, ptr, storage_of, stream_of
from ...data.cache import TensorCache
from ...data.dtype import QuantDataType
from ...data.range import QuantRange, RangeBound
from ...data.zero import ZeroPointDomain
from ...utils.common import num2str
from ...utils.config import configclass
from ..config.kernel import BaseQuantKernel, BaseQuantKernelConfig

__all__ = ["QuantGptqConfig", "QuantGptqKernel", "gptq_quantize", "hessian_accumulate", "hessian_of",
           "prepare_hessians", "clear_hessian_cache", "cholesky_inverse_upper", "gemm_f32"]

LAZY_BLOCKS = 4  # sweep blocks per far-column update
# the far updates of 2048 and more columns as two-float16-term products (``dcq_gemm_f32_two_term``; the sweep is then
# handed the scratch they need); False: three bfloat16 terms, no scratch.  Float32-GEMM accuracy either way.
TWO_TERM_UPDATES = True
SWEEP_COLUMNS = 128  # columns per sweep launch; the result does not depend on the block width (only fp order does)


@configclass
@dataclass
class QuantGptqConfig(BaseQuantKernelConfig):
    damp_percentage: float = 0.01
    block_size: int = 128
    num_inv_tries: int = 200
    hessian_block_size: int = -1

    @property
    def name(self) -> str:
        return "GPTQ"

    def build(self) -> "QuantGptqKernel":
        return QuantGptqKernel(self)

    def generate_dirnames(self, *, prefix: str = "", **kwargs) -> list[str]:
        name = f"gptq.d{num2str(self.damp_percentage)}.b{num2str(self.block_size)}"
        return [f"{prefix}.{name}" if prefix else name]


class QuantGptqKernel(BaseQuantKernel):
    def __init__(self, config: QuantGptqConfig):
        self.config = config

    def quantize(self, tensor: torch.Tensor, *, view_shape: torch.Size, quant_dtype: QuantDataType,
                 zero_domain: ZeroPointDomain | None, scale: torch.Tensor, zero: torch.Tensor, inputs: TensorCache,
                 quant_range: QuantRange | None = None, range_bound: RangeBound | None = None,
                 **kwargs) -> torch.Tensor:
        assert not tensor.requires_grad, "tensor must not require gradient."
        return gptq_quantize(tensor, view_shape=view_shape, quant_dtype=quant_dtype, zero_domain=zero_domain,
                             scale=scale, zero=zero, gptq_config=self.config, inputs=inputs, quant_range=quant_range,
                             range_bound=range_bound)


class _State(threading.local):
    """Per host thread, because independent groups of linears may be driven from one thread per HIP stream
    (``deepcompressor_amd.parallel.run_on_streams``): a workspace or memo shared across streams would race.

    workspaces: one scratch buffer per device.
    hessian: the weights fed by the same activations (q/k/v, gate/up) share one Hessian -- the most recent one is
    kept, keyed by the identity of the cached activation tensors, which the memo holds by reference: a freed
    cache whose address the allocator hands to the next layer's cache can therefore never match.
    factor: ... and therefore one act-order permutation and one inverse factor, kept for the most recent Hessian
    tensor (held by reference, so its storage cannot be recycled under the key)."""

    def __init__(self):
        self.workspaces: dict[torch.device, torch.Tensor] = {}
        self.hessian: dict = {"key": None, "refs": None, "value": None}
        self.factor: dict = {"key": None, "value": None}


_state = _State()
# Hessians computed ahead of time by ``prepare_hessians`` (any number, visible to every host thread) until
# ``clear_hessian_cache``; read-only once stored.
_prepared: dict = {}  # key -> (activation tensors held alive, Hessian, prepared ahead for the next layer)
_prepared_lock = threading.Lock()
PREPARED_LIMIT = 16  # prepared Hessians kept at most (oldest dropped first): a caller that never clears stays bounded


def clear_hessian_cache(keep_ahead: bool = False) -> None:
    """Drop this thread's memoised Hessian and inverse factor, and every prepared Hessian.  ``keep_ahead``: the
    Hessians prepared with ``ahead=True`` (the next layer's, computed beside this layer's pipelines) stay, as ordinary
    entries from now on -- the next call drops them."""
    _state.hessian["key"] = _state.hessian["refs"] = _state.hessian["value"] = None
    _state.factor["key"] = _state.factor["value"] = None
    with _prepared_lock:
        kept = {k: (refs, h, False) for k, (refs, h, ahead) in _prepared.items() if ahead} if keep_ahead else {}
        _prepared.clear()
        _prepared.update(kept)


def prepare_hessians(caches, device: torch.device, world: int = 1, round_width: int = 0,
                     profile_events: tuple | None = None, ahead: bool = False) -> list[torch.Tensor]:
    """Compute the Hessian of every activation cache in ``caches`` now, one after the other on the current stream
    (each launch has the whole chip), and keep them for ``hessian_of`` calls from any thread until
    ``clear_hessian_cache`` -- call that when the layer is done: every entry holds its cache's activation tensors
    alive (multi-GB at the calibration sizes of a language model; at most ``PREPARED_LIMIT`` entries are kept, the
    oldest go first).  The pipelines that consume the Hessians (inverse
    factor, sweeps) are latency-bound and may then run concurrently on several streams without slowing the
    matrix-core SYRKs down.

    ``world`` > 1: the calibration tokens are sharded over the processes of the default group (every rank passes the
    cache of its own shard).  H is a sum over tokens, so each rank accumulates its shard with the weight 2 / (total
    number of samples) and the partial Hessians are all-reduced: every rank ends up with the same full H (the same
    as unsharded up to the order of the float32 sums).

    ``round_width`` > 0: the SYRK takes at most that many compute units at a time, for Hessians computed while
    latency-bound kernels of other streams need a free compute unit to start on.  ``profile_events``: a pair of
    ``torch.cuda.Event(enable_timing=True)`` recorded right before / after the SYRK launches of the first cache that
    is not already prepared (a measurement hook).  ``ahead``: these are the NEXT layer's Hessians, queued while the
    current layer's pipelines run; they survive the ``clear_hessian_cache(keep_ahead=True)`` that ends the current
    layer.  Every Hessian computed here carries ``_dcq_ready``, an event recorded on the current stream right behind
    its kernels, for consumers on other streams."""
    from ...parallel import all_reduce_sum

    out = []
    for inputs in caches:
        key = _inputs_key(inputs, device)
        hit = _prepared_lookup(key, inputs)
        if hit is None:
            events, profile_events = profile_events, None
            if world > 1:
                # on the accelerator: the RCCL backend has no path for host tensors
                total = all_reduce_sum(torch.tensor([float(inputs.num_samples)], dtype=torch.float64, device=device),
                                       world)
                hit = all_reduce_sum(_compute_hessian(inputs, device, num_samples=int(total.item()),
                                                      round_width=round_width, profile_events=events), world)
            else:
                hit = _compute_hessian(inputs, device, round_width=round_width, profile_events=events)
            hit._dcq_ready = torch.cuda.Event()
            hit._dcq_ready.record()
            with _prepared_lock:
                _prepared[key] = (tuple(inputs.data), hit, ahead)
                while len(_prepared) > PREPARED_LIMIT:
                    _prepared.pop(next(iter(_prepared)))
        out.append(hit)
    return out


def _same_tensors(refs, inputs: TensorCache) -> bool:
    return refs is not None and len(refs) == len(inputs.data) and all(a is b for a, b in zip(refs, inputs.data))


def _prepared_lookup(key, inputs: TensorCache):
    with _prepared_lock:
        entry = _prepared.get(key)
    if entry is not None and _same_tensors(entry[0], inputs):
        return entry[1]
    return None


def is_prepared(inputs: TensorCache, device: torch.device) -> bool:
    """A Hessian for this cache has been computed by ``prepare_hessians`` and is still kept."""
    return _prepared_lookup(_inputs_key(inputs, device), inputs) is not None


def cached_hessian_key():
    """Key of this thread's memoised Hessian (see ``_inputs_key``), or None."""
    return _state.hessian["key"]


def _version_of(t: torch.Tensor) -> int:
    """In-place modification counter; tensors made under ``torch.inference_mode`` keep none (0 then)."""
    try:
        return t._version
    except RuntimeError:
        return 0


def _inputs_key(inputs: TensorCache, device: torch.device):
    """Hashable part of a cache's identity.  ``id`` / ``data_ptr`` only mean something while the tensor objects are
    alive, so every memo keeps the tensors themselves next to this key and compares them with ``is``
    (``_same_tensors``).  The in-place counter catches edits made outside ``torch.inference_mode``; tensors created
    under it carry no counter -- after editing such a cache in place call ``clear_hessian_cache``."""
    return (device, inputs.num_samples, inputs.channels_dim,
            tuple((id(t), t.data_ptr(), tuple(t.shape), t.dtype, _version_of(t)) for t in inputs.data))


def inputs_key(inputs: TensorCache, device: torch.device):
    """Two caches with the same key hold the same tensors and share one Hessian (valid while the tensors are alive)."""
    return _inputs_key(inputs, device)


def hessian_of(inputs: TensorCache, device: torch.device) -> torch.Tensor:
    """2 / num_samples * sum_x x^T x over the cache (float32, C x C), memoised for the last cache seen."""
    key = _inputs_key(inputs, device)
    hit = _prepared_lookup(key, inputs)
    if hit is not None:
        return hit
    _hessian_cache = _state.hessian
    if _hessian_cache["key"] != key or not _same_tensors(_hessian_cache["refs"], inputs):
        _hessian_cache["key"] = _hessian_cache["refs"] = _hessian_cache["value"] = None
        value = _compute_hessian(inputs, device)
        _hessian_cache["key"], _hessian_cache["refs"], _hessian_cache["value"] = key, tuple(inputs.data), value
    return _hessian_cache["value"]


def _compute_hessian(inputs: TensorCache, device: torch.device, num_samples: int | None = None, round_width: int = 0,
                     profile_events: tuple | None = None) -> torch.Tensor:
    hessian = None
    alpha = 2.0 / (inputs.num_samples if num_samples is None else num_samples)
    last = len(inputs.data) - 1
    for i, x in enumerate(inputs.data):
        x2 = inputs.reshape(x.view(-1, *x.shape[inputs.channels_dim:]))
        if hessian is None:
            hessian = torch.zeros((x2.shape[1], x2.shape[1]), device=device, dtype=torch.float32)
        start = profile_events[0] if profile_events is not None and i == 0 else None
        stop = profile_events[1] if profile_events is not None and i == last else None
        hessian_accumulate(hessian, x2.to(device=device, non_blocking=True), alpha, round_width=round_width,
                           start_event=start, stop_event=stop)
    if hessian is not None and num_samples is None:
        _start_diag_mean(hessian)
    return hessian


def _diag_for_damping(hessian: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """(dead channels, diagonal with the dead entries set to 1): gptq.py:199-204."""
    diag = hessian.diagonal()
    dead = diag == 0
    return dead, torch.where(dead, torch.ones_like(diag), diag)


def _start_diag_mean(hessian: torch.Tensor) -> None:
    """The damping is a fraction of the mean of the Hessian's diagonal, a host-side number: start its copy to pinned
    memory right behind the Hessian's kernels, so that the factorisation that needs it later finds it there instead
    of draining the stream for it."""
    mean = _diag_for_damping(hessian)[1].mean().reshape(1)
    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
    host.copy_(mean, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    hessian._dcq_diag_mean = (host, done)


def _diag_mean(hessian: torch.Tensor, diag: torch.Tensor) -> float:
    early = getattr(hessian, "_dcq_diag_mean", None)
    if early is None:
        return float(diag.mean())
    host, done = early
    done.synchronize()
    return float(host[0])


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    _workspaces = _state.workspaces
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        _workspaces.pop(device, None)
        ws = None
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


def release_workspaces() -> None:
    _state.workspaces.clear()


def _event_handle(event) -> C.c_void_p | None:
    """hipEvent_t of a ``torch.cuda.Event`` (torch creates it lazily on the first record: record once if needed)."""
    if event is None:
        return None
    if not event.cuda_event:
        event.record()
    return C.c_void_p(event.cuda_event)


def hessian_accumulate(hessian: torch.Tensor, x: torch.Tensor, alpha: float, *, round_width: int = 0,
                       start_event=None, stop_event=None) -> None:
    """hessian[C][C] += alpha * x^T x for x[N][C] on the same GPU: 16-bit activations take the bf16 / f16
    matrix-core SYRK, float32 activations the float32 matrix-core GEMM (x^T is the transposed-A operand).
    ``round_width``, ``start_event`` / ``stop_event`` (``torch.cuda.Event`` with timing): the scheduling hint and the
    measurement hook of ``dcq_hessian_accumulate_ex`` (16-bit activations only)."""
    assert x.ndim == 2 and hessian.shape == (x.shape[1], x.shape[1]) and hessian.dtype == torch.float32
    x = x.contiguous()
    if x.dtype == torch.float32:
        if x.shape[0] % 32:  # the GEMM steps K in multiples of 32: zero rows add nothing
            x = torch.cat([x, x.new_zeros(32 - x.shape[0] % 32, x.shape[1])], dim=0)
        gemm_f32(x, x, hessian, alpha=float(alpha), beta=1.0, a_transposed=True)
        return
    T, Cn = x.shape
    nbytes = lib().dcq_hessian_workspace_bytes(T, Cn)
    ws = _workspace(x.device, nbytes)
    check(lib().dcq_hessian_accumulate_ex(ptr(x), storage_of(x.dtype), T, Cn, float(alpha), ptr(hessian), ptr(ws),
                                          ws.numel(), int(round_width), _event_handle(start_event),
                                          _event_handle(stop_event), stream_of(x.device)), "dcq_hessian_accumulate_ex")


def gemm_f32(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, *, alpha: float = 1.0, beta: float = 0.0,
             a_transposed: bool = False, two_term: bool = False) -> None:
    """c = beta * c + alpha * a @ b on the float32 matrix-core path (``a`` holds a^T when ``a_transposed``).
    Operands are 2-D float32 views with unit stride along their last dimension.  ``two_term`` forces the path that
    multiplies two float16 terms per operand, scaled per row of ``a`` and per column of ``b``
    (``dcq_gemm_f32_two_term``), which the inverse factor's large products take."""
    for t in (a, b, c):
        assert t.dtype == torch.float32 and t.ndim == 2 and t.stride(1) == 1, "float32 row-major views expected"
    K, N = b.shape
    M = a.shape[1] if a_transposed else a.shape[0]
    assert (a.shape[0] if a_transposed else a.shape[1]) == K and c.shape == (M, N), "shape mismatch"
    if M == 0 or N == 0:
        return
    raw = lambda t: C.c_void_p(t.data_ptr())  # strided views: the leading dimensions carry the layout
    engine._require_gpu(c)
    if two_term:
        ws = _workspace(c.device, lib().dcq_gemm_f32_two_term_workspace_bytes(M, N, K))
        check(lib().dcq_gemm_f32_two_term(raw(a), a.stride(0), int(a_transposed), raw(b), b.stride(0), raw(c), c.stride(0),
                                          M, N, K, float(alpha), float(beta), ptr(ws), ws.numel(), stream_of(c.device)),
              "dcq_gemm_f32_two_term")
        return
    check(lib().dcq_gemm_f32(raw(a), a.stride(0), int(a_transposed), raw(b), b.stride(0), raw(c), c.stride(0), M, N, K,
                             float(alpha), float(beta), stream_of(c.device)), "dcq_gemm_f32")


def _enqueue_factor(hessian: torch.Tensor, perm: torch.Tensor | None, diag_add: float):
    """One attempt of ``dcq_cholesky_inverse_upper`` with the given damping, enqueued: (U, info) -- ``info`` (device
    int32) is 0 once the kernels have run and every pivot was positive."""
    n = hessian.shape[0]
    assert hessian.shape == (n, n) and hessian.dtype == torch.float32 and hessian.is_contiguous()
    dev = hessian.device
    u = torch.empty((n, n), device=dev, dtype=torch.float32)
    info = torch.empty(1, device=dev, dtype=torch.int32)
    ws = _workspace(dev, lib().dcq_cholesky_workspace_bytes(n))
    check(lib().dcq_cholesky_inverse_upper(ptr(hessian), n, ptr(perm), float(diag_add), ptr(u), ptr(ws), ws.numel(),
                                           ptr(info), stream_of(dev)), "dcq_cholesky_inverse_upper")
    return u, info


def _damping(damp: float, diag_mean: float, attempt: int) -> float:
    """gptq.py:209-233: ``damp * mean(diag)``, a tenth more of it on every failed factorisation."""
    return damp * diag_mean * (1.0 + 0.1 * attempt)


def cholesky_inverse_upper(hessian: torch.Tensor, perm: torch.Tensor | None, diag_mean: float, damp: float,
                           num_tries: int):
    """U (upper) with U^T U = S^-1 for S = P H' P^T + damping, H' = ``hessian`` with dead (zero) diagonal
    entries set to 1, P the act-order gather ``perm``.  Damping starts at ``damp * diag_mean`` and grows by a tenth
    of that on each failed factorisation (gptq.py:209-233).  Returns (U, tries); ``hessian`` is left untouched.

    The reference factors S = L L^T, forms S^-1 from L and factors that again.  The same U comes out of ONE
    factorisation: with J the index reversal, J S J = R^T R (R upper) gives S = V V^T with V = J R^T J upper
    triangular, hence S^-1 = (V^-1)^T (V^-1) and, by uniqueness of the Cholesky factor, U = V^-1 -- a
    triangular inverse instead of a symmetric inverse plus a second factorisation (half the flops, and no
    second loss of accuracy).  Both steps run in ``dcq_cholesky_inverse_upper``.
    """
    tries, done, u = 0, False, None
    while not done and tries < num_tries:
        u, info = _enqueue_factor(hessian, perm, _damping(damp, diag_mean, tries))
        tries += 1
        done = int(info.item()) == 0
    assert done, "Hessian could not be factorised"
    engine.assert_finite(u, "Inverse of Hessian matrix", inf_first=True)
    return u, tries


def _inverse_factor_of(hessian: torch.Tensor, config: QuantGptqConfig, attempt: int = 0):
    """(dead channels, act-order permutation, its inverse, U, info) of a Hessian (gptq.py:199-233), memoised for the
    most recent Hessian tensor: weights that share their inputs share them.  The factorisation is only enqueued here:
    ``info`` is the device flag of attempt number ``attempt`` while nobody has looked at it yet (the caller queues the
    work that uses U behind it, reads the flag afterwards and reports through ``_factor_verified``), None once it
    has been seen to be zero."""
    key = (_version_of(hessian), config.damp_percentage, config.num_inv_tries)
    _factor_cache = _state.factor
    hit = _factor_cache["key"]
    if hit is not None and hit[0] is hessian and hit[1] == key:
        value = _factor_cache["value"]
        if value[4] is None or value[5] == attempt:  # a verified factor serves every caller, whatever it took
            return value[:5]
    _factor_cache["key"] = _factor_cache["value"] = None
    dead, diag = _diag_for_damping(hessian)
    perm = torch.argsort(diag, descending=True)
    inverse_perm = torch.argsort(perm)
    U, info = _enqueue_factor(hessian, perm, _damping(config.damp_percentage, _diag_mean(hessian, diag), attempt))
    _factor_cache["key"], _factor_cache["value"] = (hessian, key), (dead, perm, inverse_perm, U, info, attempt)
    return _factor_cache["value"][:5]


def _factor_verified(hessian: torch.Tensor) -> None:
    """The memoised factor's flag has been read as zero: later users need not look again."""
    _factor_cache = _state.factor
    hit = _factor_cache["key"]
    if hit is not None and hit[0] is hessian:
        dead, perm, inverse_perm, U, _, attempt = _factor_cache["value"]
        _factor_cache["value"] = (dead, perm, inverse_perm, U, None, attempt)


@torch.no_grad()
def gptq_quantize(tensor: torch.Tensor, *, view_shape: torch.Size, quant_dtype: QuantDataType,
                  zero_domain: ZeroPointDomain | None, scale: torch.Tensor, zero: torch.Tensor,
                  gptq_config: QuantGptqConfig, inputs: TensorCache, quant_range: QuantRange | None = None,
                  range_bound: RangeBound | None = None, hessian: torch.Tensor | None = None) -> torch.Tensor:
    """Quantized values of ``tensor`` (float32, float16 or bfloat16; left untouched: the float32 working copy is made
    by the act-order gather) in the shape ``view_shape``.

    ``hessian`` (optional, not in the reference signature) lets callers that quantize several weights fed by the
    same activations (q/k/v, gate/up) pass one precomputed 2/n * X^T X instead of recomputing it per weight.
    """
    engine._require_gpu(tensor)
    view = tensor.view(view_shape)
    view_shape = view.shape
    assert len(view_shape) >= 4 and len(view_shape) % 2 == 0, "view_shape is (#g0, gs0, #g1, gs1, ...)"
    # (#g0, gs0, #g1, gs1, #g2, gs2, ...) -> rows (#g0 gs0) x columns (#g1 #g2 ... gs1 gs2 ...): the columns of one
    # quantization group side by side (gptq.py:168-178); for a linear's (#g0, gs0, #g1, gs1) nothing moves
    n_view = len(view_shape)
    n_rg, rg_size = view_shape[0], view_shape[1]
    n_cg, cg_size = view_shape[2::2].numel(), view_shape[3::2].numel()
    R, Cn = n_rg * rg_size, n_cg * cg_size
    dev = tensor.device
    W = view.permute(0, 1, *range(2, n_view, 2), *range(3, n_view, 2)).reshape(R, Cn) if n_view > 4 else view.reshape(R, Cn)
    assert W.dtype in (torch.float32, torch.float16, torch.bfloat16), "the weight must be float32 / float16 / bfloat16"
    # ---- Hessian (shared, never modified), dead channels, act-order permutation, damping, inverse factor
    hessian = hessian_of(inputs, dev) if hessian is None else hessian
    assert hessian.shape == (Cn, Cn), "activation channels do not match the weight's input channels"
    # ---- sweep set-up
    qr = QuantRange.construct(quant_dtype, has_zero_point=zero_domain is not None, quant_range=quant_range)
    step = engine.make_step(quant_dtype=quant_dtype, zero_domain=zero_domain, tensor_quant_range=qr,
                            quant_range=quant_range, scale_dtype=torch.float32, zero_dtype=torch.float32,
                            range_bound=range_bound, q_span=1.0, device=dev)
    keep = [v if i % 2 == 0 else 1 for i, v in enumerate(view_shape)]
    s2 = scale.to(torch.float32).expand(keep).reshape(n_rg, n_cg).contiguous()
    zero_is_number = isinstance(zero, (int, float)) or zero.numel() == 1
    z2, z_scalar = None, 0.0
    if zero_domain is not None:
        if zero_is_number:
            z_scalar = float(zero)
        else:
            z2 = zero.to(torch.float32).reshape(n_rg, n_cg).contiguous()
    # Lazy updates at two levels (same sums as gptq.py:269, regrouped): the errors of LAZY_BLOCKS consecutive
    # sweep blocks are collected side by side; columns inside the group are updated after every block (K = 128),
    # the columns right of the group once per group with one K = 128 * LAZY_BLOCKS product, which reads and
    # writes the far part of W LAZY_BLOCKS times less often.
    group_cols = SWEEP_COLUMNS * LAZY_BLOCKS
    err = torch.empty((R, group_cols), device=dev, dtype=torch.float32)
    # scratch of the far updates' two-term products (magnitudes and term planes of the errors and of U's rows)
    scratch = (torch.empty(lib().dcq_gptq_sweep_workspace_bytes(R, Cn, LAZY_BLOCKS), device=dev, dtype=torch.uint8)
               if TWO_TERM_UPDATES else None)
    # The factorisation is enqueued, not waited for: the sweep goes into the stream right behind it, and only then is
    # the factorisation's flag read.  A failed attempt (a non-positive pivot: rare, U is garbage then and so is this
    # sweep's output) is repeated with more damping, as the reference does (gptq.py:209-233).
    attempt = 0
    while True:
        dead, perm, inverse_perm, U, info = _inverse_factor_of(hessian, gptq_config, attempt)
        Wp = engine.gather_columns(W, perm, zero_columns=dead)  # dead channels zeroed, act-order (gptq.py:199-208)
        Q = torch.empty_like(Wp)  # the sweep writes every row and column
        check(lib().dcq_gptq_sweep(C.byref(step), ptr(Wp), Cn, R, Cn, ptr(U), ptr(s2), ptr(z2), z_scalar, rg_size,
                                   cg_size, n_cg, ptr(perm), ptr(Q), ptr(err), LAZY_BLOCKS, ptr(scratch), 0 if scratch is None else scratch.numel(),
                                   stream_of(dev)),
              "dcq_gptq_sweep")
        if info is None:
            break
        if int(info.item()) == 0:
            _factor_verified(hessian)
            engine.assert_finite(U, "Inverse of Hessian matrix", inf_first=True)
            break
        attempt += 1
        assert attempt < gptq_config.num_inv_tries, "Hessian could not be factorised"
    Q = engine.gather_columns(Q, inverse_perm)
    engine.assert_finite(Q, "GPTQ Quantized tensor")
    if n_view == 4:
        return Q.reshape(n_rg, rg_size, n_cg, cg_size).contiguous()
    # back to (#g0, gs0, #g1, gs1, #g2, gs2, ...) (gptq.py:272-281)
    half = n_view // 2
    order = [0, 1] + [d for i in range(1, half) for d in (1 + i, half + i)]
    return Q.reshape(view_shape[:2] + view_shape[2::2] + view_shape[3::2]).permute(order).reshape(view_shape).contiguous()
