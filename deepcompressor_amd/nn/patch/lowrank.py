"""Low-rank side branch of SVDQuant (reference: deepcompressor/nn/patch/lowrank.py:12-103).

``LowRankBranch(in, out, rank, weight=W)`` holds ``a = Vh[:rank]`` and ``b = U[:, :rank] * S[:rank]`` of the float64
SVD of ``W``, cast to ``W``'s dtype, so ``b @ a`` is the best rank-``rank`` approximation of ``W``.

The reference takes a full float64 SVD for that (2.4 s for a 3072 x 3072 weight on this GPU, once per calibration
iteration).  Only the leading ``rank`` singular triplets are kept, so here they come from the Gram matrix of the
smaller side: ``G = W^T W`` (or ``W W^T``) on the float64 matrix cores (``dcq_gemm_f64``), the ``rank`` leading
eigenpairs of G by a Chebyshev-filtered block iteration on the same matrix-core products (``_leading_eigenpairs``;
a dense symmetric eigensolver on all n pairs took 0.09 s at 3072), and one more float64 product ``U S = W V``.  In
float64 the squared condition number of the Gram route costs nothing that survives the cast to the weight's dtype.  Singular vectors are unique up to sign only, so
parity with the reference is stated on ``b @ a`` and on the residual, not on ``a`` / ``b`` separately.
"""

from __future__ import annotations

import torch
import torch.nn as nn

__all__ = ["LowRankBranch", "AccumBranchHook", "leading_singular_triplets", "add_low_rank_branches"]


def _split_k_f64(a: torch.Tensor, b: torch.Tensor, *, a_transposed: bool = False, splits: int | None = None) -> torch.Tensor:
    """``a @ b`` (``a^T @ b`` when ``a_transposed``: ``a`` is [K][M]) in float64 for products that are long in K and
    narrow in N (a Gram matrix times a block of vectors, a block's own Gram matrix): the K range is cut into slabs that
    run as one batched ``dcq_gemm_f64`` launch -- enough workgroups to fill the chip where one per 64 x 64 output tile
    would leave most compute units idle and each workgroup alone with its load latency -- and the slabs' partial
    products are added up."""
    import ctypes as C

    from ..._lib import check, lib, stream_of

    a, b = a.contiguous(), b.contiguous()
    assert a.dtype == b.dtype == torch.float64 and a.ndim == b.ndim == 2
    k, n = b.shape
    m = a.shape[1] if a_transposed else a.shape[0]
    assert (a.shape[0] if a_transposed else a.shape[1]) == k
    tiles = -(-m // 64) * -(-n // 64)
    if splits is None:  # about three workgroups per compute unit, slabs of at least 64 and a multiple of 16 long
        splits = max(1, min(768 // max(1, tiles), k // 64))
    while splits > 1 and (k % splits or (k // splits) % 16):
        splits -= 1
    kc = k // splits
    out = torch.empty((splits, m, n), dtype=torch.float64, device=b.device)
    raw = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    lda = a.shape[1]
    a_batch = kc * lda if a_transposed else kc  # the next slab of a k-major / row-major first operand
    check(lib().dcq_gemm_f64(raw(a), 3, lda, int(a_transposed), a_batch, raw(b), 3, n, kc * n, raw(out), 3, n, m * n,
                             m, n, kc, splits, stream_of(b.device)), "dcq_gemm_f64")
    return out[0] if splits == 1 else out.sum(dim=0)


# Below this order the dense symmetric eigensolver is as fast as the block iteration's own small ones
DIRECT_EIGH_ORDER = 768
BLOCK = 128  # vectors iterated together (the wanted ones plus a margin that sets the convergence rate)
FILTER_PASSES = 2  # filter + orthonormalization passes between two Rayleigh-Ritz steps


GRAM_KERNEL_ORDER = 128  # largest k x k Gram matrix that ``dcq_gram_inverse_factor_f64`` keeps in LDS


def _gram_inverse_factor(gram: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """(R^-1, info) with R^T R = (gram + gram^T) / 2, R upper triangular: ``cholesky_ex`` + ``solve_triangular``
    against the identity in one launch (float64, order <= GRAM_KERNEL_ORDER)."""
    import ctypes as C

    from ..._lib import check, lib, stream_of

    raw = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    k = gram.shape[0]
    assert gram.dtype == torch.float64 and gram.shape == (k, k) and gram.is_contiguous()
    r_inv = torch.empty_like(gram)
    info = torch.empty(1, dtype=torch.int32, device=gram.device)
    check(lib().dcq_gram_inverse_factor_f64(raw(gram), k, k, raw(r_inv), k, raw(info), stream_of(gram.device)),
          "dcq_gram_inverse_factor_f64")
    return r_inv, info[0]


def _orthonormalize(x: torch.Tensor) -> torch.Tensor:
    """Columns of x [n][k] orthonormalized by two rounds of Cholesky QR (x <- x R^-1 with R^T R = x^T x): everything
    large is a float64 matrix-core product, the k x k factorisations are the only small dense routines."""
    k = x.shape[1]
    eye = None if k <= GRAM_KERNEL_ORDER else torch.eye(k, dtype=x.dtype, device=x.device)
    given, failed = x, None
    for _ in range(2):
        gram = _split_k_f64(x, x, a_transposed=True)
        if eye is None:  # symmetrized, factored and inverted by one workgroup with the matrix in LDS
            r_inv, info = _gram_inverse_factor(gram)
        else:
            gram = (gram + gram.t()) * 0.5
            r, info = torch.linalg.cholesky_ex(gram, upper=True)
            r_inv = torch.linalg.solve_triangular(r, eye, upper=True)
        failed = info if failed is None else failed | info
        x = _split_k_f64(x, r_inv, splits=1)
    # both rounds are queued before the host looks at the factorisations' flags (one read instead of two waits)
    if int(failed.item()) != 0:  # columns (numerically) dependent: the slow but unconditional route
        return torch.linalg.qr(given).Q
    return x


def _chebyshev_filter(gram: torch.Tensor, x: torch.Tensor, degree: int, cut: float, top: float) -> torch.Tensor:
    """p(G) x for the Chebyshev polynomial p of the given degree that stays within [-1, 1] on [0, cut] (the unwanted
    part of the spectrum of the positive semi-definite G) and grows as fast as a polynomial can above it, scaled so
    that p(top) = 1: the three-term recurrence y_next = 2 s' / e (G - c) y - s s' x, one product with G per degree.
    The shift c is taken out of G's diagonal for the duration, so a step is the product and one scaled addition."""
    e = c = cut / 2.0
    diag = gram.diagonal()
    diag.sub_(c)
    try:
        sigma = e / (top - c)
        sigma1 = sigma
        y = _split_k_f64(gram, x).mul_(sigma1 / e)
        for _ in range(1, degree):
            sigma2 = 1.0 / (2.0 / sigma1 - sigma)
            y_next = _split_k_f64(gram, y).mul_(2.0 * sigma2 / e).add_(x, alpha=-sigma * sigma2)
            x, y, sigma = y, y_next, sigma2
    finally:
        diag.add_(c)
    return y


def _leading_eigenpairs(gram: torch.Tensor, rank: int, tol: float = 1e-9, max_rounds: int = 16):
    """The ``rank`` largest eigenvalues (descending) and their eigenvectors [n][rank] of a symmetric positive
    semi-definite float64 matrix (which is used as scratch and restored only approximately: pass a copy to keep it).

    Only ``rank`` of the n pairs are wanted (32 of 3072 for a FLUX.1 linear), so instead of a dense eigensolver this
    runs Chebyshev-filtered subspace iteration with locking.  A block of BLOCK vectors is rotated to the eigenvectors
    of its k x k projection (Rayleigh-Ritz); leading pairs whose residual |G v - lambda v| is below ``tol`` *
    lambda_max are locked and taken out of the matrix (G <- G - v lambda v^T, so they fall into the damped part of the
    spectrum and cannot swamp the block again); the rest of the block is multiplied by a polynomial of G that damps the
    spectrum below a cut -- the block's smallest Ritz value -- and amplifies it above (``_chebyshev_filter``, its degree
    from the gap the Ritz values show between the last wanted pair and the cut, bounded so that the largest remaining
    eigenvalue does not outgrow the others by more than Cholesky QR can take), and orthonormalized again.  All n-sized
    work is float64 matrix-core products (``dcq_gemm_f64``); the k x k Cholesky and symmetric eigenproblems are the
    small dense routines that remain.  Matrices of low order, and ones that have not converged after ``max_rounds``
    rounds, take the dense eigensolver."""
    import math

    n = gram.shape[0]
    k = BLOCK if rank <= BLOCK // 2 else min(n, 2 * rank + 64)

    def dense(matrix):
        values, vectors = torch.linalg.eigh(matrix)  # ascending
        return values[-rank:].flip(0), vectors[:, -rank:].flip(1).contiguous()

    if n <= max(DIRECT_EIGH_ORDER, 4 * k):
        return dense(gram)
    original = gram
    gram = gram.clone()  # deflated in place as pairs are locked
    gen = torch.Generator(device=gram.device).manual_seed(0x5eed)
    fresh = lambda cols: torch.randn((n, cols), dtype=torch.float64, device=gram.device, generator=gen)  # noqa: E731
    # two steps of plain power iteration: enough for eigenvalues far above the rest to show up in the first projection
    x = _orthonormalize(fresh(k))
    for _ in range(2):
        x = _orthonormalize(_split_k_f64(gram, x))
    locked_values: list[torch.Tensor] = []
    locked_vectors: list[torch.Tensor] = []
    need = rank
    scale = None
    for _ in range(max_rounds):
        gx = _split_k_f64(gram, x)
        h = _split_k_f64(x, gx, a_transposed=True)
        theta, y = torch.linalg.eigh((h + h.t()) * 0.5)  # ascending
        x, gx = _split_k_f64(x, y, splits=1), _split_k_f64(gx, y, splits=1)
        resid = (gx - x * theta).norm(dim=0)
        stats = torch.cat([theta, resid]).cpu()  # one host read per round
        theta_h, resid_h = stats[:k].tolist(), stats[k:].tolist()
        if scale is None:
            scale = theta_h[-1]  # lambda_max of the matrix (to Ritz accuracy): the residual bar's unit
            if not scale > 0.0:
                return dense(original)
        # the leading run of converged pairs
        done = 0
        while done < need and resid_h[k - 1 - done] <= tol * scale:
            done += 1
        if done:
            vals, vecs = theta[k - done:].flip(0), x[:, k - done:].flip(1).contiguous()
            locked_values.append(vals)
            locked_vectors.append(vecs)
            need -= done
            if need == 0:
                break
            gram.sub_(_split_k_f64(vecs * vals, vecs.t().contiguous(), splits=1))
            x = torch.cat([x[:, :k - done], fresh(done)], dim=1)
        active = k - done
        top, cut = theta_h[active - 1], theta_h[0]
        wanted = theta_h[active - need] if need <= active else cut
        if not (cut > 0.0 and top > cut):
            return dense(original)
        # degree: amplify the last wanted pair ~1e5 over the cut per filter, but the largest remaining eigenvalue no
        # more than ~1e7 (the block's condition number must stay inside what Cholesky QR takes)
        gain = lambda lam: math.acosh(1.0 + 2.0 * max(lam - cut, 1e-7 * top) / cut)  # noqa: E731
        degree = int(max(2, min(40, math.ceil(math.log(2e5) / gain(wanted)), math.floor(math.log(2e7) / gain(top)))))
        for _ in range(FILTER_PASSES):
            x = _orthonormalize(_chebyshev_filter(gram, x, degree, cut, top))
    else:
        return dense(original)
    values, vectors = torch.cat(locked_values), torch.cat(locked_vectors, dim=1)
    if len(locked_vectors) > 1:  # groups locked in different rounds are orthogonal to ~tol only: tidy up
        vectors = _orthonormalize(vectors)
    # locked pairs came out in descending order round by round; verify them against the matrix as given
    resid = (_split_k_f64(original, vectors) - vectors * values).norm(dim=0).max()
    if not float(resid) <= 100.0 * tol * scale:
        return dense(original)
    return values, vectors


def leading_singular_triplets(weight: torch.Tensor, rank: int) -> tuple[torch.Tensor, torch.Tensor]:
    """``(U[:, :rank] * S[:rank], Vh[:rank])`` of the float64 SVD of a 2-D GPU tensor, both float64."""
    from ...utils.math.hadamard import gemm_f64

    out_features, in_features = weight.shape
    rank = min(rank, out_features, in_features)
    weight = weight.contiguous()
    if in_features <= out_features:
        gram = gemm_f64(weight, weight, torch.float64, a_transposed=True)  # W^T W
        _, v = _leading_eigenpairs(gram, rank)  # [in, rank], leading first
        return gemm_f64(weight, v, torch.float64), v.t().contiguous()  # U S = W V
    gram = gemm_f64(weight, weight.t().contiguous(), torch.float64)  # W W^T
    values, u = _leading_eigenpairs(gram, rank)  # [out, rank]
    sigma = values.clamp(min=0).sqrt()
    svh = gemm_f64(u, weight, torch.float64, a_transposed=True)  # U^T W = S Vh
    # a vanishing singular value leaves its direction undetermined (and its column of U S zero): any unit row will do
    safe = torch.where(sigma > 0, sigma, torch.ones_like(sigma))
    return u * sigma, svh / safe[:, None]


class AccumBranchHook:
    """Forward hook: ``output += branch(input)`` (the reference's ``AccumBranchHook``, utils/hooks/branch.py)."""

    def __init__(self, branch: nn.Module | None, input_packager=None, output_packager=None):
        self.branch, self.input_packager, self.output_packager = branch, input_packager, output_packager
        self._handles: list = []
        self._inputs = None

    def _pre(self, module, args, kwargs):
        if self.input_packager is not None:
            tensors = self.input_packager.unpack(module, args, kwargs)
            assert len(tensors) == 1, "a branch takes one input tensor"
            self._inputs = next(iter(tensors.values()))
        else:
            self._inputs = args[0]
        return None

    def _post(self, module, args, kwargs, output):
        x, self._inputs = self._inputs, None
        if self.branch is None:
            return None
        extra = self.branch(x)
        if extra is None:
            return None
        if isinstance(output, torch.Tensor):
            return output + extra
        return (output[0] + extra, *output[1:])

    def register(self, module: nn.Module | list[nn.Module]) -> "AccumBranchHook":
        for m in [module] if isinstance(module, nn.Module) else list(module):
            self._handles.append(m.register_forward_pre_hook(self._pre, with_kwargs=True))
            self._handles.append(m.register_forward_hook(self._post, with_kwargs=True))
        return self

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles.clear()


class LowRankBranch(nn.Module):
    def __init__(self, in_features: int, out_features: int, rank: int, alpha: float = 1.0,
                 weight: torch.Tensor | None = None):
        super().__init__()
        self.in_features, self.out_features, self.rank, self.alpha = in_features, out_features, rank, alpha
        if rank == 0:
            self.a, self.b = None, None
        elif rank < 0:  # a full-rank "branch": the weight itself
            self.a, self.b = nn.Linear(in_features, out_features, bias=False), nn.Identity()
        else:
            self.a, self.b = nn.Linear(in_features, rank, bias=False), nn.Linear(rank, out_features, bias=False)
        self.reset_parameters(weight)

    @torch.no_grad()
    def reset_parameters(self, weight: torch.Tensor | None = None) -> None:
        """Without ``weight``: the usual LoRA start (random ``a``, zero ``b``; a zero matrix for the full-rank form).
        With it: the leading singular triplets of ``weight`` (lowrank.py:30-59 of the reference)."""
        if self.rank == 0:
            return
        if weight is None:
            if self.rank > 0:
                nn.init.kaiming_uniform_(self.a.weight)
                nn.init.zeros_(self.b.weight)
            else:
                nn.init.zeros_(self.a.weight)
            return
        assert weight.ndim < 2 or weight.shape[2:].numel() == 1, "LinearLoRAHook only supports 2D input tensor"
        matrix = weight.view(weight.shape[0], -1)
        assert matrix.shape[1] == self.in_features, "Input features size mismatch"
        assert matrix.shape[0] == self.out_features, "Output features size mismatch"
        self.to(device=matrix.device, dtype=matrix.dtype)
        if self.rank < 0:
            self.a.weight.data.copy_(matrix)
            return
        factors = dict(zip(("U * S", "V^T"), leading_singular_triplets(matrix, self.rank)))
        for probe, what in ((torch.isnan, "NaN"), (torch.isinf, "Inf")):
            for name, value in factors.items():
                assert not probe(value).any(), f"{what} in {name}"
        self.b.weight.data.copy_(factors["U * S"].to(matrix.dtype))
        self.a.weight.data.copy_(factors["V^T"].to(matrix.dtype))

    def get_effective_weight(self) -> torch.Tensor | None:
        if self.rank == 0:
            return None
        if self.rank < 0:
            return self.a.weight
        return self.b.weight @ self.a.weight

    def forward(self, input: torch.Tensor) -> torch.Tensor | None:
        if self.a is None:
            return None
        if input.ndim <= 3:
            return self.alpha * self.b(self.a(input))
        assert input.ndim == 4
        assert input.shape[-1] != self.in_features
        assert input.shape[1] == self.in_features
        B, Cin, H, W = input.shape  # channels-first images: to [B, H * W, C] and back
        out = self.alpha * self.b(self.a(input.permute(0, 2, 3, 1).reshape(B, H * W, Cin)))
        return out.reshape(B, H, W, -1).permute(0, 3, 1, 2)

    def as_hook(self, input_packager=None, output_packager=None) -> AccumBranchHook:
        return AccumBranchHook(self, input_packager=input_packager, output_packager=output_packager)


@torch.no_grad()
def add_low_rank_branches(modules: list[nn.Module], state_dict: dict[str, torch.Tensor], rank: int) -> list[LowRankBranch]:
    """Put one stored entry of ``branch.pt`` to work (reference: app/diffusion/quant/weight.py:117-144).

    ``state_dict`` holds ``a.weight`` [rank, in] and ``b.weight`` [sum of the modules' output features, rank] of the
    branch calibrated for ``modules`` together (modules that share their input, e.g. q / k / v, share ``a``).  Every
    module gets a branch with the shared ``a`` and its own rows of ``b``; the branch's effective weight is taken out
    of the module's weight, and the branch is hooked behind the module so that the sum of both is unchanged.
    Returns the branches in module order."""
    first = modules[0].weight
    in_features = first.shape[1]
    shared = LowRankBranch(in_features=in_features, out_features=sum(m.weight.shape[0] for m in modules), rank=rank)
    shared.to(device=first.device, dtype=first.dtype)
    shared.load_state_dict(state_dict)
    if len(modules) == 1:
        first.data.sub_(shared.get_effective_weight().view(first.shape))
        shared.as_hook().register(modules[0])
        return [shared]
    branches, row = [], 0
    for module in modules:
        weight = module.weight
        rows = weight.shape[0]
        branch = LowRankBranch(in_features=in_features, out_features=rows, rank=rank)
        branch.a = shared.a
        branch.b.to(device=weight.device, dtype=weight.dtype)
        branch.b.weight.copy_(shared.b.weight[row:row + rows])
        row += rows
        weight.data.sub_(branch.get_effective_weight().view(weight.shape))
        branch.as_hook().register(module)
        branches.append(branch)
    return branches
