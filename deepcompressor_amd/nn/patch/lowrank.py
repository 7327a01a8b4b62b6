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

__all__ = ["LowRankBranch", "AccumBranchHook", "leading_singular_triplets"]


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
        if weight is None:
            if self.rank < 0:
                nn.init.zeros_(self.a.weight)
            elif self.rank > 0:
                nn.init.kaiming_uniform_(self.a.weight)
                nn.init.zeros_(self.b.weight)
            return
        if weight.ndim >= 2:
            assert weight.shape[2:].numel() == 1, "LinearLoRAHook only supports 2D input tensor"
        weight = weight.view(weight.shape[0], -1)
        self.to(device=weight.device, dtype=weight.dtype)
        out_features, in_features = weight.shape
        assert self.in_features == in_features, "Input features size mismatch"
        assert self.out_features == out_features, "Output features size mismatch"
        if self.rank < 0:
            self.a.weight.data.copy_(weight)
        elif self.rank > 0:
            us, vh = leading_singular_triplets(weight, self.rank)
            assert not us.isnan().any(), "NaN in U * S"
            assert not vh.isnan().any(), "NaN in V^T"
            assert not us.isinf().any(), "Inf in U * S"
            assert not vh.isinf().any(), "Inf in V^T"
            self.a.weight.data.copy_(vh.to(weight.dtype))
            self.b.weight.data.copy_(us.to(weight.dtype))

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
