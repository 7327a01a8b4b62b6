"""Import the read-only reference package in the build container (never on the GPU box).

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/make_golden.py`` to produce the fixtures under
``tests/golden/`` and by the optional ``reference``-marked CPU tests.  Nothing in the product
package imports this file.

The reference (``/root/reference/deepcompressor``) needs three things this image lacks:

* ``typing.Self`` (python 3.10)                       -> taken from ``typing_extensions``;
* the PyPI package ``omniconfig`` (decorator + argparse metadata only, no numerics)
                                                      -> an empty module with an identity ``configclass``;
* its CUDA extension ``deepcompressor.csrc.load._C`` (needs nvcc, ``csrc/load.py:12-33``)
                                                      -> ``round_to_nearest_in_codebook_cuda`` restated in torch
                                                         from ``csrc/quantize/quantize.cu:69-99``.

Because the third item is a restatement and not the reference's own binary, every fixture whose
value depends on floating-point codebook rounding is marked ``"codebook_via_restatement": true`` in
its metadata; integer-dtype fixtures never touch it.
"""

from __future__ import annotations

import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get("DEEPCOMPRESSOR_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "deepcompressor"))


def _codebook_round(tensor, codebook, inplace=False, bnb=False):
    """Torch restatement of ``nearest_neighbor`` (``csrc/quantize/quantize.cu:69-99``).

    lo = last index with codebook[lo] <= x (0 if none), hi = min(lo + 1, C - 1),
    result = codebook[lo] if x + x < codebook[lo] + codebook[hi] else codebook[hi].
    """
    import torch

    x = tensor.contiguous().reshape(-1)
    c = codebook.contiguous()
    lo = (torch.searchsorted(c, x, right=True) - 1).clamp_(min=0)
    hi = (lo + 1).clamp_(max=c.numel() - 1)
    return torch.where(x + x < c[lo] + c[hi], c[lo], c[hi]).view_as(tensor)


def load_reference():
    """Return the imported ``deepcompressor`` package of the reference (CPU tensors only)."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    if "deepcompressor" in sys.modules and getattr(sys.modules["deepcompressor"], "_dcq_oracle", False):
        return sys.modules["deepcompressor"]
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if not hasattr(typing, "Self"):
        import typing_extensions

        typing.Self = typing_extensions.Self
    if "omniconfig" not in sys.modules:
        oc = types.ModuleType("omniconfig")
        oc.configclass = lambda cls: cls
        oc.ARGPARSE_KWARGS, oc.ARGPARSE_ARGS = "argparse_kwargs", "argparse_args"
        oc.Arguments = oc.ConfigParser = object
        oc.ADD_PREFIX_BOOL_FIELDS = lambda *a, **k: (lambda parser: None)
        oc.COLLECT_PREFIX_BOOL_FIELDS = lambda *a, **k: []
        sys.modules["omniconfig"] = oc
    sys.path.insert(0, REFERENCE_ROOT)
    import deepcompressor  # noqa: E402  (only imports .version)

    csrc = types.ModuleType("deepcompressor.csrc")
    csrc.__path__ = []
    load = types.ModuleType("deepcompressor.csrc.load")
    load._C = types.SimpleNamespace(round_to_nearest_in_codebook_cuda=_codebook_round)
    sys.modules["deepcompressor.csrc"], sys.modules["deepcompressor.csrc.load"] = csrc, load
    deepcompressor._dcq_oracle = True
    return deepcompressor
