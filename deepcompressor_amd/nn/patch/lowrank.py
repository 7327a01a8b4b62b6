"""Low-rank side branch of SVDQuant (reference: deepcompressor/nn/patch/lowrank.py:12-103).

``LowRankBranch(in, out, rank, weight=W)`` holds ``a = Vh[:rank]`` and ``b = U[:, :rank] * S[:rank]`` of the float64
SVD of ``W``, cast to ``W``'s dtype, so ``b @ a`` is the best rank-``rank`` approximation of ``W``.

The reference takes a full float64 SVD for that (2.4 s for a 3072 x 3072 weight on this GPU, once per calibration
iteration).  Only the leading ``rank`` singular triplets are kept, so here they come from the Gram matrix of the
smaller side: ``G = W^T W`` (or ``W W^T``) on the float64 matrix cores (``dcq_gemm_f64``), its symmetric
eigendecomposition (the vendor's LAPACK through ``torch.linalg.eigh`` -- the one delegated routine of this row,
0.09 s at 3072), and one more float64 product ``U S = W V``.  In float64 the squared condition number of the Gram
route costs nothing that survives the cast to the weight's dtype.  Singular vectors are unique up to sign only, so
parity with the reference is stated on ``b @ a`` and on the residual, not on ``a`` / ``b`` separately.
"""

from __future__ import annotations

import torch
import torch.nn as nn

__all__ = ["LowRankBranch", "AccumBranchHook", "leading_singular_triplets", "add_low_rank_branches"]


def leading_singular_triplets(weight: torch.Tensor, rank: int) -> tuple[torch.Tensor, torch.Tensor]:
    """``(U[:, :rank] * S[:rank], Vh[:rank])`` of the float64 SVD of a 2-D GPU tensor, both float64."""
    from ...utils.math.hadamard import gemm_f64

    out_features, in_features = weight.shape
    rank = min(rank, out_features, in_features)
    weight = weight.contiguous()
    if in_features <= out_features:
        gram = gemm_f64(weight, weight, torch.float64, a_transposed=True)  # W^T W
        _, vectors = torch.linalg.eigh(gram)  # ascending eigenvalues
        v = vectors[:, -rank:].flip(1).contiguous()  # [in, rank], leading first
        return gemm_f64(weight, v, torch.float64), v.t().contiguous()  # U S = W V
    gram = gemm_f64(weight, weight.t().contiguous(), torch.float64)  # W W^T
    values, vectors = torch.linalg.eigh(gram)
    u = vectors[:, -rank:].flip(1).contiguous()  # [out, rank]
    sigma = values[-rank:].flip(0).clamp_(min=0).sqrt_()
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
