"""Rotation of weights (reference: deepcompressor/calib/rotate.py:24-245).

Same functions and signatures as the reference.  Every product that the reference evaluates as
``matmul(x.to(float64), y).to(dtype)`` is one ``dcq_gemm_f64`` launch: operands are widened to float64 while they
are staged, accumulated on the float64 matrix cores and rounded double -> float -> dtype on the way out, so no
float64 copy of a weight is ever materialised.
"""

from __future__ import annotations

import typing as tp

import torch
import torch.nn as nn

from ..utils.math import HadamardMatrix, hardmard_transform, random_hadamard_matrix
from ..utils.math.hadamard import gemm_f64

__all__ = ["rotate_in_channels", "rotate_out_channels", "hadamard_in_channels", "get_rotation_matrix",
           "transform_rms_norm_and_linear", "transform_layer_norm_to_rms_norm", "transform_norm_and_linear",
           "RMSNorm", "HadamardTransformHook"]


class RMSNorm(nn.Module):
    """x / rms(x) * weight with the statistics in float32 (rotate.py:24-39)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        dtype = hidden_states.dtype
        h = hidden_states.to(torch.float32)
        h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * h.to(dtype)


class HadamardTransformHook:
    """Forward pre-hook that applies the Hadamard transform to a module's inputs (rotate.py:42-63)."""

    def __init__(self, rhs: torch.Tensor, lhs: torch.Tensor, lhs_k: int, scaled: bool = True, packager=None):
        self.rhs, self.lhs, self.lhs_k, self.scaled, self.input_packager = rhs, lhs, lhs_k, scaled, packager
        self._handles: list = []

    def _transform(self, x: torch.Tensor) -> torch.Tensor:
        return hardmard_transform(x, hadamard_rhs=self.rhs, hadamard_lhs=self.lhs, lhs_k=self.lhs_k, scaled=self.scaled)

    def pre_forward(self, module: nn.Module, input_args: tuple, input_kwargs: dict):
        if self.input_packager is None:
            return (self._transform(input_args[0]), *input_args[1:]), input_kwargs
        tensors = self.input_packager.unpack(module, input_args, input_kwargs)
        tensors = {k: self._transform(x) for k, x in tensors.items()}
        return self.input_packager.repack(tensors, module, input_args, input_kwargs)

    def register(self, module: nn.Module | tp.Iterable[nn.Module]) -> "HadamardTransformHook":
        for m in [module] if isinstance(module, nn.Module) else list(module):
            self._handles.append(m.register_forward_pre_hook(self.pre_forward, with_kwargs=True))
        return self

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles.clear()


def rotate_in_channels(weight: nn.Parameter, /, *, rotation: torch.Tensor) -> None:
    """weight <- (weight.view(-1, n) @ rotation) with float64 accumulation (rotate.py:66-73)."""
    shape, dtype = weight.shape, weight.dtype
    w2 = weight.data.reshape(-1, rotation.shape[0])
    weight.data = gemm_f64(w2, rotation.to(device=weight.device, dtype=torch.float64), dtype).view(shape)


def rotate_out_channels(weight: nn.Parameter, /, *, rotation: torch.Tensor, bias: nn.Parameter | None = None) -> None:
    """Per head of ``rotation.shape[0]`` output channels: weight_h <- rotation^T @ weight_h (rotate.py:76-95)."""
    shape, dtype = weight.shape, weight.dtype
    head_channels = rotation.shape[0]
    num_heads = shape[0] // head_channels
    rot = rotation.to(device=weight.device, dtype=torch.float64)
    # rotation is [k = head channel in][m = head channel out]: exactly the transposed-A layout of the GEMM
    weight.data = gemm_f64(rot, weight.data.reshape(num_heads, head_channels, -1), dtype, a_transposed=True).view(shape)
    if bias is not None:
        bias.data = gemm_f64(rot, bias.data.reshape(num_heads, head_channels, -1), dtype, a_transposed=True).view(-1)


def hadamard_in_channels(modules: tp.Iterable[nn.Module], packager=None, dtype: torch.dtype | None = None,
                         device: torch.device | str | None = None) -> None:
    """Hadamard-transform the input channels of linear modules and hook the same transform onto their inputs
    (rotate.py:98-117)."""
    for module in modules:
        if not isinstance(module, nn.Linear):
            raise NotImplementedError(f"Module {module} not supported!")
        in_channels = module.in_features
        device, dtype = device or module.weight.device, dtype or module.weight.dtype
        rhs64, lhs64, k = HadamardMatrix.get(in_channels, scale=True, dtype=torch.float64)
        w_dtype = module.weight.dtype
        w = module.weight.data
        rhs_k = in_channels // k
        # float64 throughout, one rounding to the weight dtype at the end (the reference transforms weight.to(float64))
        y = gemm_f64(w.reshape(-1, rhs_k), rhs64.to(w.device), torch.float64 if k > 1 else w_dtype)
        if k > 1:
            y = gemm_f64(lhs64.to(w.device), y.view(-1, k, rhs_k), w_dtype)
        module.weight.data = y.view(w.shape).to(device=device)
        rhs, lhs, k = HadamardMatrix.get(in_channels, scale=True, dtype=dtype, device=device)
        HadamardTransformHook(rhs=rhs, lhs=lhs, lhs_k=k, packager=packager).register(module)


def get_rotation_matrix(num_channels: int, random: bool = True, compatible: bool = True) -> torch.Tensor:
    """A float64 orthogonal matrix: a randomly signed Hadamard matrix, or the fixed one (rotate.py:120-133)."""
    if random:
        return random_hadamard_matrix(num_channels)
    rhs, lhs, k = HadamardMatrix.get(num_channels, scale=False)
    rhs = rhs.to(dtype=torch.float64)
    if k == 1:
        rotation = rhs
    elif compatible:  # the matrix that hardmard_transform applies
        rotation = torch.kron(lhs.T.contiguous().to(dtype=torch.float64), rhs)
    else:
        rotation = torch.kron(rhs, lhs.to(dtype=torch.float64))
    return rotation.mul_(1.0 / torch.tensor(num_channels, dtype=torch.float64).sqrt())


def transform_rms_norm_and_linear(norm: nn.Module, next_modules: tp.Iterable[nn.Linear]) -> None:
    """Fold the norm's weight (and bias) into the linears that consume its output (rotate.py:136-162)."""
    ln_w = norm.weight.data.to(dtype=torch.float64)
    norm.weight.data = torch.ones_like(norm.weight.data)
    ln_b = None
    if getattr(norm, "bias", None) is not None:
        ln_b = norm.bias.data.to(dtype=torch.float64)
        norm.bias = None
    for linear in next_modules:
        assert isinstance(linear, nn.Linear)
        dtype = linear.weight.dtype
        fc_w = linear.weight.data.to(dtype=torch.float64)
        ln_w = ln_w.to(fc_w.device)
        linear.weight.data = (fc_w * ln_w).to(dtype=dtype)
        if ln_b is not None:
            ln_b = ln_b.to(fc_w.device)
            if linear.bias is None:
                linear.bias = nn.Parameter(torch.zeros(linear.out_features, dtype=dtype, device=linear.weight.device))
            shift = gemm_f64(fc_w, ln_b.view(-1, 1), torch.float64).view(-1)
            linear.bias.data = (linear.bias.data.to(dtype=torch.float64) + shift).to(dtype=dtype)


def transform_layer_norm_to_rms_norm(parent: nn.Module, norm_name: str, prev_modules: tp.Iterable[nn.Module],
                                     prev_out_channels_dims: int | tp.Iterable[int] = 0) -> None:
    """LayerNorm (bias already folded away) -> RMSNorm: the mean subtraction moves into the producers' weights
    (rotate.py:165-216)."""
    for name in norm_name.split(".")[:-1]:
        parent = getattr(parent, name)
    norm_name = norm_name.split(".")[-1]
    norm = getattr(parent, norm_name)
    assert isinstance(norm, nn.LayerNorm)
    assert len(norm.normalized_shape) == 1, f"LayerNorm's #dims must be 1, got {len(norm.normalized_shape)}"
    assert norm.bias is None, "LayerNorm's bias must be None, please call `transform_rms_norm_and_linear` in advance"
    prev_modules = list(prev_modules)
    assert len(prev_modules) > 0, "No previous modules found"
    if isinstance(prev_out_channels_dims, int):
        prev_out_channels_dims = [prev_out_channels_dims] * len(prev_modules)
    for module, dim in zip(prev_modules, prev_out_channels_dims, strict=True):
        if isinstance(module, nn.LayerNorm):
            module.bias = None
            continue
        if isinstance(module, nn.Linear):
            assert dim == 0, "Linear module's output channels dimension is 0"
        elif isinstance(module, nn.Embedding):
            assert dim == 1, "Embedding module's output channels dimension is 1"
        dtype = module.weight.dtype
        w = module.weight.data.to(dtype=torch.float64)
        module.weight.data = w.sub_(w.mean(dim=dim, keepdim=True)).to(dtype=dtype)
        if getattr(module, "bias", None) is not None:
            b = module.bias.data.to(dtype=torch.float64)
            module.bias.data = b.sub_(b.mean()).to(dtype=dtype)
    rms = RMSNorm(hidden_size=norm.normalized_shape[0], eps=norm.eps)
    rms.weight.data = norm.weight.data
    setattr(parent, norm_name, rms)


def transform_norm_and_linear(parent: nn.Module, norm_name: str, next_modules: tp.Iterable[nn.Linear],
                              prev_modules: tp.Iterable[nn.Module] | None = None,
                              prev_out_channels_dims: int | tp.Iterable[int] = 0) -> None:
    """Both steps above for the norm ``parent.<norm_name>`` (rotate.py:219-245)."""
    holder = parent
    for name in norm_name.split(".")[:-1]:
        holder = getattr(holder, name)
    norm = getattr(holder, norm_name.split(".")[-1])
    transform_rms_norm_and_linear(norm, next_modules)
    if isinstance(norm, nn.LayerNorm):
        transform_layer_norm_to_rms_norm(parent, norm_name, prev_modules, prev_out_channels_dims)
