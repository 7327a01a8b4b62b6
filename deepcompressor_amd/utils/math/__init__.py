from .functional import is_pow2, root_
from .hadamard import HadamardMatrix, hardmard_transform, random_hadamard_matrix

__all__ = ["is_pow2", "root_", "HadamardMatrix", "hardmard_transform", "random_hadamard_matrix"]
