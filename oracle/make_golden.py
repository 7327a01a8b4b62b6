"""Generate the golden fixtures under ``tests/golden/`` by running the *reference* on seeded inputs.

Runs only in the build container (needs ``/root/reference``); the GPU box uses the committed
``.npz`` files.  Every fixture stores its inputs (as float32 arrays of values exactly representable
in the nominal dtype) and the reference's outputs, so tests can replay them against the oracle and
against the HIP path without the reference present.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TORCH_DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}

# name -> (quantizer kwargs in plain python, input shape)
FAKE_QUANT_CASES = {
    # Llama W4 g128 configs (examples/llm/configs/gptq.yaml, awq.yaml)
    "sint4_g128_f16s": dict(dtype="sint4", zero=None, groups=((1, 128),), sdt=("f16",), shape=(64, 512)),
    "uint4_post_g128_f16s": dict(dtype="uint4", zero="PostScale", groups=((1, 128),), sdt=("f16",), shape=(64, 512)),
    "uint4_pre_g128_f16s": dict(dtype="uint4", zero="PreScale", groups=((1, 128),), sdt=("f16",), shape=(64, 512)),
    "sint4_pre_g128_defs": dict(dtype="sint4", zero="PreScale", groups=((1, 128),), sdt=(None,), shape=(32, 256)),
    # SVDQuant int4 g64 (examples/diffusion/configs/svdquant/int4.yaml)
    "sint4_g64_defs": dict(dtype="sint4", zero=None, groups=((1, 64),), sdt=(None,), shape=(48, 384)),
    # W8A8 per-channel / per-tensor
    "sint8_gchn_f16s": dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",), shape=(64, 512)),
    "sint8_tnsr_f16s": dict(dtype="sint8", zero=None, groups=((-1, -1),), sdt=("f16",), shape=(32, 256)),
    "uint8_post_gchn_defs": dict(dtype="uint8", zero="PostScale", groups=((1, -1),), sdt=(None,), shape=(32, 320)),
    # NVFP4 (examples/diffusion/configs/svdquant/nvfp4.yaml): two-level scale, fp8 group scale
    "nvfp4_g16": dict(dtype="sfp4_e2m1_all", zero=None, groups=((-1, -1), (1, 16)), sdt=(None, "sfp8_e4m3_nan"),
                      shape=(32, 256)),
    "sfp4_g32_f16s": dict(dtype="sfp4_e2m1_all", zero=None, groups=((1, 32),), sdt=("f16",), shape=(32, 256)),
    "sfp8_e4m3_gchn": dict(dtype="sfp8_e4m3_nan", zero=None, groups=((1, -1),), sdt=("f16",), shape=(16, 256)),
    # two-level integer scales: per-channel f16 then g32 int8 scale inside one step
    "sint4_gchn_g32_s8": dict(dtype="sint4", zero=None, groups=((1, -1), (1, 32)), sdt=("f16", "sint8"),
                              shape=(32, 256)),
    # QoQ W4A8 (examples/llm/configs/qoq-g128.yaml): progressive sint8 per-channel -> uint4 g128, int8 scales
    "qoq_g128": dict(dtype="uint4", zero="PostScale", groups=((1, -1), (1, 128)), sdt=("f16", "sint8"),
                     inter=("sint8",), levels=(0,), shape=(64, 512)),
    "qoq_g128_pre": dict(dtype="uint4", zero="PreScale", groups=((1, -1), (1, 128)), sdt=("f16", "sint8"),
                         inter=("sint8",), levels=(0,), shape=(32, 256)),
    # QoQ per-channel (qoq-gchn.yaml): plain sint4 per-channel
    "sint4_gchn_f16s": dict(dtype="sint4", zero=None, groups=((1, -1),), sdt=("f16",), shape=(64, 512)),
    # odd group geometry: 2-row groups, and a 4-D (conv-like) weight
    "sint4_2x64": dict(dtype="sint4", zero=None, groups=((2, 64),), sdt=("f16",), shape=(32, 256)),
    "uint4_post_conv": dict(dtype="uint4", zero="PostScale", groups=((1, 16, -1),), sdt=("f16",), shape=(16, 64, 3, 3)),
}
DYN_CASES = {
    # (case, dynamic range spec)
    "sint4_g128_f16s@ratio0.9": ("sint4_g128_f16s", dict(ratio=0.9)),
    "uint4_post_g128_f16s@ratio0.85": ("uint4_post_g128_f16s", dict(ratio=0.85)),
    "uint4_pre_g128_f16s@ratio0.7": ("uint4_pre_g128_f16s", dict(ratio=0.7)),
}


def _mk_input(shape, dt: str, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(*shape, generator=g)
    # heavy tail on 0.5 % of the entries so clipping matters, plus one all-zero group and one constant row
    mask = torch.rand(*shape, generator=g) < 0.005
    w = torch.where(mask, w * 6, w) * 0.02
    flat = w.view(shape[0], -1)
    flat[0, : min(128, flat.shape[1])] = 0
    if shape[0] > 1:
        flat[1] = 0.0123
    return w.to(TORCH_DT[dt])


def _build_quantizer(ref, case, default_dtype=None):
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.zero import ZeroPointDomain
    from deepcompressor.quantizer import Quantizer
    from deepcompressor.quantizer.config.base import ProgressiveQuantizerConfig, QuantizerConfig

    def sd(x):
        if x is None:
            return None
        return TORCH_DT[x] if x in TORCH_DT else QuantDataType.from_str(x)

    kw = dict(
        dtype=QuantDataType.from_str(case["dtype"]),
        zero_point=None if case["zero"] is None else ZeroPointDomain[case["zero"]],
        group_shapes=case["groups"],
        scale_dtypes=tuple(sd(x) for x in case["sdt"]),
    )
    if "inter" in case:
        cfg = ProgressiveQuantizerConfig(
            **kw,
            intermediate_dtypes=tuple(QuantDataType.from_str(x) for x in case["inter"]),
            intermediate_levels=case["levels"],
        )
    else:
        cfg = QuantizerConfig(**kw)
    return Quantizer(cfg, default_dtype=default_dtype)


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()


def gen_fake_quant(ref) -> None:
    from deepcompressor.data.range import DynamicRange

    meta = {}
    arrays = {}
    jobs = [(name, name, None) for name in FAKE_QUANT_CASES] + [(k, v[0], v[1]) for k, v in DYN_CASES.items()]
    for key, cname, dyn in jobs:
        case = FAKE_QUANT_CASES[cname]
        for dt in ("bf16", "f16", "f32"):
            if dyn is not None and dt != "bf16":
                continue
            w = _mk_input(case["shape"], dt, seed=hash((cname, dt)) % 1000 if False else len(cname) * 7 + len(dt))
            q = _build_quantizer(ref, case)
            kwargs = {}
            if dyn is not None:
                kwargs["dynamic_range"] = DynamicRange(ratio=dyn["ratio"])
            codes = q.quantize(w, return_with_dequant=False, return_with_quant=True, **kwargs)
            deq = q.quantize(w, return_with_dequant=True, return_with_quant=False, **kwargs)
            tag = f"{key}|{dt}"
            arrays[f"{tag}|x"] = _np(w)
            arrays[f"{tag}|codes"] = _np(codes.qdata)
            arrays[f"{tag}|dequant"] = _np(deq.data)
            arrays[f"{tag}|scale"] = _np(codes.scale.data)
            sd = codes.scale.state_dict("s")
            for lvl, leaf in enumerate(sd.values()):
                arrays[f"{tag}|leaf{lvl}"] = _np(leaf)
            arrays[f"{tag}|zero"] = _np(codes.zero)
            meta[tag] = dict(case=cname, in_dtype=dt, dyn=dyn, view_shape=list(codes.view_shape), n_leaves=len(sd),
                             codebook_via_restatement=case["dtype"].startswith("sfp")
                             or any(isinstance(s, str) and s.startswith("sfp") for s in case["sdt"]))
    np.savez_compressed(os.path.join(OUT, "fake_quant.npz"), **arrays)
    with open(os.path.join(OUT, "fake_quant.json"), "w") as f:
        json.dump(dict(cases={k: {kk: vv for kk, vv in v.items() if kk != "shape"} | {"shape": list(v["shape"])}
                              for k, v in FAKE_QUANT_CASES.items()}, entries=meta), f, indent=1)


def gen_range_search(ref) -> None:
    import torch.nn as nn
    from deepcompressor.calib.config import (
        DynamicRangeCalibConfig,
        SearchBasedCalibGranularity,
        SearchBasedCalibObjective,
        SearchBasedCalibStrategy,
    )
    from deepcompressor.calib.range import DynamicRangeCalibrator
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.common import TensorType
    from deepcompressor.data.utils.reshape import LinearReshapeFn

    arrays, meta = {}, {}
    plan = [
        ("sint4_g128_f16s", "bf16", 2.4, 100), ("sint4_g128_f16s", "f16", 2.4, 100),
        ("sint4_g128_f16s", "f32", 2.4, 100), ("sint4_g128_f16s", "bf16", 2, 80),
        ("uint4_post_g128_f16s", "bf16", 2.4, 100), ("uint4_post_g128_f16s", "f16", 2.4, 100),
        ("uint4_pre_g128_f16s", "bf16", 2.4, 100), ("sint4_g64_defs", "bf16", 2.4, 100),
        ("sint4_g64_defs", "f16", 2, 80), ("sint4_gchn_f16s", "bf16", 2.4, 100), ("sint8_gchn_f16s", "f16", 2, 80),
        ("sfp4_g32_f16s", "bf16", 2.4, 100), ("uint4_post_g128_f16s", "bf16", 3, 40),
    ]
    for cname, dt, degree, grids in plan:
        case = FAKE_QUANT_CASES[cname]
        w = _mk_input(case["shape"], dt, seed=1000 + len(cname) + len(dt))
        q = _build_quantizer(ref, case)
        cfg = DynamicRangeCalibConfig(
            degree=degree, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0, num_grids=grids,
        )
        cal = DynamicRangeCalibrator(TensorType.Weights, cfg, static=True, quantizer=q)
        errors = []
        orig_tell = cal._tell

        def tell(error, _orig=orig_tell, _errors=errors):
            _errors.append(_np(error[0]).copy())
            _orig(error)

        cal._tell = tell
        dummy = TensorsCache(TensorCache([torch.zeros(1, 8, w.shape[-1])], channels_dim=-1, reshape=LinearReshapeFn()))
        best = cal.calibrate(x_wgts=[nn.Parameter(w, requires_grad=False)], eval_inputs=dummy)
        tag = f"{cname}|{dt}|d{degree}|g{grids}"
        arrays[f"{tag}|x"] = _np(w)
        arrays[f"{tag}|max"] = _np(best.max)
        if best.min is not None:
            arrays[f"{tag}|min"] = _np(best.min)
        arrays[f"{tag}|ratio"] = _np(cal.best_range.ratio)
        arrays[f"{tag}|best_error"] = _np(cal.best_error)
        arrays[f"{tag}|errors"] = np.stack(errors)
        arrays[f"{tag}|ratios"] = np.asarray([r for it in cal.ratios for r in it], dtype=np.float64)
        meta[tag] = dict(case=cname, in_dtype=dt, degree=degree, num_grids=grids,
                         codebook_via_restatement=case["dtype"].startswith("sfp"))
    np.savez_compressed(os.path.join(OUT, "range_search.npz"), **arrays)
    with open(os.path.join(OUT, "range_search.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_tables(ref) -> None:
    """pow tables over every finite non-negative 16-bit input, ATen sum-order samples, codebooks, protective ranges."""
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.range import ProtectiveQuantRange
    from deepcompressor.data.zero import ZeroPointDomain

    arrays = {}
    for dt, end in (("bf16", 0x7F80), ("f16", 0x7C00)):
        bits = torch.arange(0, end, dtype=torch.int32).to(torch.int16)
        a = bits.view(TORCH_DT[dt])
        for degree in (2.4, 2, 3, 1.5, 2.5, 4, 1, 2.2):
            p = a.clone().pow_(degree)
            arrays[f"pow|{dt}|{degree}"] = p.view(torch.int16).numpy().astype(np.uint16)
    g = torch.Generator().manual_seed(7)
    for dt in ("bf16", "f16", "f32"):
        for S in (8, 16, 32, 64, 96, 128, 200, 256, 512, 4096):
            x = (torch.rand(192 if S <= 512 else 24, S, generator=g) ** 3 * 1e-3).to(TORCH_DT[dt])
            arrays[f"sum|{dt}|{S}|x"] = _np(x)
            arrays[f"sum|{dt}|{S}|y"] = _np(x.sum(dim=-1))
    for name in ("sfp4_e2m1_all", "sfp8_e4m3_nan", "sfp8_e5m2_inf", "ufp4_e2m2_all", "sfp6_e3m2_all", "sfp6_e2m3_all",
                 "sint4", "uint4", "smag4"):
        arrays[f"codebook|{name}"] = QuantDataType.from_str(name).get_codebook().values.numpy().astype(np.float32)
    prot = {}
    for outer, inner, zd in (("sint8", "uint4", "PostScale"), ("sint8", "uint4", "PreScale"),
                             ("sint8", "sint4", "PostScale"), ("sint8", "sint4", "PreScale"), ("sint8", "sint4", None)):
        r = ProtectiveQuantRange.construct(QuantDataType.from_str(outer), QuantDataType.from_str(inner),
                                           None if zd is None else ZeroPointDomain[zd])
        prot[f"{outer}|{inner}|{zd}"] = [r.min, r.max]
    # dtype scalar properties
    props = {}
    for name in ("sint4", "uint4", "sint8", "uint8", "smag4", "sfp4_e2m1_all", "sfp8_e4m3_nan", "sfp8_e5m2_inf",
                 "sfp6_e3m2_all", "uexp8_all", "sexp4_nan", "ufp4_e2m2_all"):
        d = QuantDataType.from_str(name)
        props[name] = dict(min=d.min_value, max=d.max_value, max_exp=d.max_exponent_value, min_exp=d.min_exponent_value,
                           is_exp=d.is_exponent, is_fp=d.is_float_point, signed=d.signed, bits=d.total_bits)
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **arrays)
    with open(os.path.join(OUT, "tables.json"), "w") as f:
        json.dump(dict(protective=prot, dtype_props=props), f, indent=1)


def gen_gptq(ref) -> None:
    from deepcompressor.data.cache import TensorCache
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.quantizer.kernel.gptq import QuantGptqConfig

    arrays, meta = {}, {}
    for cname, dt, R, C, T in (("uint4_post_g128_f16s", "bf16", 96, 512, 1024), ("sint4_g128_f16s", "f16", 64, 256, 512),
                               ("sint4_g64_defs", "bf16", 64, 256, 512)):
        g = torch.Generator().manual_seed(R + C)
        w = (torch.randn(R, C, generator=g) * 0.02).to(TORCH_DT[dt])
        gains = torch.exp(torch.randn(C, generator=g))
        x = (torch.randn(2, T // 2, C, generator=g) * gains).to(TORCH_DT[dt])
        x[..., 5] = 0  # a dead input channel
        case = FAKE_QUANT_CASES[cname]
        q = _build_quantizer(ref, case)
        cache = TensorCache([x], channels_dim=-1, reshape=LinearReshapeFn(), num_cached=1, num_total=1, num_samples=T)
        kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
        qt = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True)
        dq = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=True, return_with_quant=False)
        rtn = q.quantize(w, return_with_dequant=True, return_with_quant=True)
        tag = f"{cname}|{dt}|{R}x{C}|T{T}"
        arrays[f"{tag}|w"] = _np(w)
        arrays[f"{tag}|x"] = _np(x)
        arrays[f"{tag}|codes"] = _np(qt.qdata)
        arrays[f"{tag}|dequant"] = _np(dq.data)
        arrays[f"{tag}|rtn_dequant"] = _np(rtn.data)
        arrays[f"{tag}|scale"] = _np(qt.scale.data)
        arrays[f"{tag}|zero"] = _np(qt.zero)
        meta[tag] = dict(case=cname, in_dtype=dt, R=R, C=C, T=T, damp=0.01, block=128, num_samples=T)
    np.savez_compressed(os.path.join(OUT, "gptq.npz"), **arrays)
    with open(os.path.join(OUT, "gptq.json"), "w") as f:
        json.dump(meta, f, indent=1)
    gen_gptq_conv(ref)


def gen_gptq_conv(ref) -> None:
    """GPTQ on a convolution weight, whose view has more than two pairs of dimensions: the reference moves the
    columns of a quantization group side by side (quantizer/kernel/gptq.py:168-178) and back (:272-281).  Two group
    geometries: whole receptive fields of 4 channels (nothing moves) and one kernel row of all channels per group (the
    columns are re-ordered, while the Hessian stays in the activations' unfolded order, as the reference has it)."""
    from deepcompressor.data.cache import TensorCache
    from deepcompressor.data.utils.reshape import ConvInputReshapeFn
    from deepcompressor.quantizer.kernel.gptq import QuantGptqConfig

    arrays, meta = {}, {}
    O, Cin, K = 24, 8, 3
    for tag, groups in (("fields", ((1, 4, -1),)), ("kernel_rows", ((1, -1, 1, -1),))):
        case = dict(dtype="uint4", zero="PostScale", groups=groups, sdt=("f16",), shape=(O, Cin, K, K))
        g = torch.Generator().manual_seed(len(tag))
        w = (torch.randn(O, Cin, K, K, generator=g) * 0.05).to(torch.bfloat16)
        gains = torch.exp(torch.randn(1, Cin, 1, 1, generator=g) * 0.7)
        x = (torch.randn(4, Cin, 10, 10, generator=g) * gains).to(torch.bfloat16)
        reshape = ConvInputReshapeFn((K, K), (1, 1), (1, 1), (1, 1))
        q = _build_quantizer(ref, case)
        cache = TensorCache([x], channels_dim=1, reshape=reshape, num_cached=1, num_total=1, num_samples=4)
        kern = QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512)
        qt = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=False, return_with_quant=True)
        dq = q.quantize(w, kernel=kern, inputs=cache, return_with_dequant=True, return_with_quant=False)
        arrays[f"{tag}|w"], arrays[f"{tag}|x"] = _np(w), _np(x)
        arrays[f"{tag}|codes"], arrays[f"{tag}|dequant"] = _np(qt.qdata), _np(dq.data)
        arrays[f"{tag}|scale"], arrays[f"{tag}|zero"] = _np(qt.scale.data), _np(qt.zero)
        meta[tag] = dict(case=case, view_shape=list(qt.view_shape), num_samples=4)
    np.savez_compressed(os.path.join(OUT, "gptq_conv.npz"), **arrays)
    with open(os.path.join(OUT, "gptq_conv.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_rotate(ref) -> None:
    """Hadamard tables, rotation matrices and rotated weights from the reference's own functions
    (deepcompressor/calib/rotate.py, utils/math/hadamard.py) on small seeded inputs."""
    import torch.nn as nn
    from deepcompressor.calib import rotate as rrot
    from deepcompressor.utils.math.hadamard import HadamardMatrix, hardmard_transform, random_hadamard_matrix

    arrays, meta = {}, {}
    for k in (12, 20, 28, 36, 40, 52, 60, 108, 140, 156, 172):
        h = HadamardMatrix._get_hadamard_k(k).numpy()
        arrays[f"hadamard|{k}"] = np.packbits(h > 0)
    g = torch.Generator().manual_seed(77)
    cases = [("n64_bf16", 48, 64, "bf16"), ("n96_f16", 40, 96, "f16"), ("n224_bf16", 24, 224, "bf16"),
             ("n512_f32", 16, 512, "f32")]
    for tag, rows, n, dt in cases:
        w = (torch.randn(rows, n, generator=g) * 0.05).to(TORCH_DT[dt])
        arrays[f"{tag}|w"] = _np(w)
        fixed = rrot.get_rotation_matrix(n, random=False)
        torch.manual_seed(1000 + n)
        rand = rrot.get_rotation_matrix(n, random=True)
        arrays[f"{tag}|rotation_fixed"] = fixed.numpy()
        arrays[f"{tag}|rotation_random"] = rand.numpy()
        for name, rot in (("fixed", fixed), ("random", rand)):
            p = nn.Parameter(w.clone(), requires_grad=False)
            rrot.rotate_in_channels(p, rotation=rot)
            arrays[f"{tag}|in_{name}"] = _np(p.data)
        # hadamard_in_channels: weight part only (hardmard_transform in float64, then the cast)
        rhs, lhs, k = HadamardMatrix.get(n, scale=True, dtype=torch.float64)
        arrays[f"{tag}|hadamard_in"] = _np(hardmard_transform(w.to(torch.float64), rhs, lhs, k, scaled=True).to(w.dtype))
        # the transform in the tensor's own dtype, unscaled matrices (activation-side use)
        rhs_u, lhs_u, k = HadamardMatrix.get(n, scale=False, dtype=torch.float64)
        arrays[f"{tag}|hadamard_unscaled_f64"] = hardmard_transform(w.to(torch.float64), rhs_u, lhs_u, k).numpy()
        meta[tag] = dict(rows=rows, n=n, dtype=dt, k=k, seed_random=1000 + n)
    # output-channel rotation per head, with bias
    for tag, heads, hc, cols, dt in (("out_bf16", 4, 16, 72, "bf16"), ("out_f16", 3, 12, 40, "f16")):
        w = (torch.randn(heads * hc, cols, generator=g) * 0.05).to(TORCH_DT[dt])
        b = (torch.randn(heads * hc, generator=g) * 0.1).to(TORCH_DT[dt])
        torch.manual_seed(500 + hc)
        rot = rrot.get_rotation_matrix(hc, random=True)
        pw, pb = nn.Parameter(w.clone(), requires_grad=False), nn.Parameter(b.clone(), requires_grad=False)
        rrot.rotate_out_channels(pw, rotation=rot, bias=pb)
        arrays[f"{tag}|w"], arrays[f"{tag}|b"], arrays[f"{tag}|rotation"] = _np(w), _np(b), rot.numpy()
        arrays[f"{tag}|w_out"], arrays[f"{tag}|b_out"] = _np(pw.data), _np(pb.data)
        meta[tag] = dict(heads=heads, head_channels=hc, cols=cols, dtype=dt)
    # norm folding
    ln = nn.LayerNorm(64, bias=True)
    ln.weight.data = torch.randn(64, generator=g).abs() + 0.5
    ln.bias.data = torch.randn(64, generator=g) * 0.1
    fc = nn.Linear(64, 32, bias=False).to(torch.bfloat16)
    fc.weight.data = (torch.randn(32, 64, generator=g) * 0.05).to(torch.bfloat16)
    arrays["norm|ln_w"], arrays["norm|ln_b"], arrays["norm|fc_w"] = _np(ln.weight.data), _np(ln.bias.data), _np(fc.weight.data)
    rrot.transform_rms_norm_and_linear(ln, [fc])
    arrays["norm|fc_w_out"], arrays["norm|fc_b_out"] = _np(fc.weight.data), _np(fc.bias.data)
    np.savez_compressed(os.path.join(OUT, "rotate.npz"), **arrays)
    with open(os.path.join(OUT, "rotate.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_smooth(ref) -> None:
    """Spans, candidate lists, scales, applied scales and a small search from the reference's smooth module
    (deepcompressor/calib/smooth.py, calib/metric.py, calib/config/smooth.py)."""
    import torch.nn as nn
    from deepcompressor.calib import smooth as rs
    from deepcompressor.calib.config import SearchBasedCalibGranularity, SearchBasedCalibObjective, SearchBasedCalibStrategy
    from deepcompressor.calib.config.smooth import SmoothCalibConfig, SmoothSpanMode
    from deepcompressor.calib.metric import ChannelMetric
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.utils.reshape import LinearReshapeFn

    arrays, meta = {}, {"pairs": {}, "dirnames": {}}
    # candidate lists
    for alpha in (0.5, 0, -1, -2, -3):
        for beta in (0.5, 0, -1, -2, -3):
            cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=alpha, beta=beta,
                                    num_grids=5, strategy=SearchBasedCalibStrategy.GridSearch)
            meta["pairs"][f"{alpha}|{beta}"] = [list(p) for p in cfg.get_alpha_beta_pairs()]
            meta["dirnames"][f"{alpha}|{beta}"] = cfg.generate_dirnames(prefix="smooth")
    for alpha, beta in ((0.3, 0.7), (0.5, -1), (-1, 0.25), (0.5, 0)):
        cfg = SmoothCalibConfig(spans=[("AbsMax", "RootMeanSquare")], alpha=alpha, beta=beta)
        meta["pairs"][f"manual|{alpha}|{beta}"] = [list(p) for p in cfg.get_alpha_beta_pairs()]
        meta["dirnames"][f"manual|{alpha}|{beta}"] = cfg.generate_dirnames()
    # spans
    g = torch.Generator().manual_seed(99)
    for dt in ("bf16", "f16"):
        xs = [(torch.randn(n, 96, generator=g) * torch.exp(torch.randn(96, generator=g))).to(TORCH_DT[dt]) for n in (40, 24, 8)]
        xs[0][:, 5] = 0
        ws = [(torch.randn(r, 96, generator=g) * 0.05).to(TORCH_DT[dt]) for r in (64, 32)]
        conv = (torch.randn(6, 96, 3, 3, generator=g) * 0.1).to(TORCH_DT[dt])
        for i, x in enumerate(xs):
            arrays[f"span|{dt}|x{i}"] = _np(x)
        for i, w in enumerate(ws):
            arrays[f"span|{dt}|w{i}"] = _np(w)
        arrays[f"span|{dt}|conv"] = _np(conv)
        for mode in ("abs_max", "abs_mean", "root_mean_square"):
            fn = getattr(ChannelMetric, mode)
            arrays[f"span|{dt}|x|{mode}"] = fn(xs, 96, [1, -1], dtype=torch.float32).numpy()
            arrays[f"span|{dt}|w|{mode}"] = fn(ws, 96, [1, -1], dtype=torch.float32).numpy()
            arrays[f"span|{dt}|conv|{mode}"] = fn(conv, 96, [1, -1], dtype=torch.float32).numpy()
        a_span = arrays[f"span|{dt}|x|abs_max"]
        b_span = arrays[f"span|{dt}|w|abs_max"]
        for alpha, beta in ((0.5, 0.5), (0.3, 0.7), (0.85, 0), (0, 0.4), (1.0, 1.0)):
            s = rs.get_smooth_scale(alpha_base=torch.from_numpy(a_span.copy()), beta_base=torch.from_numpy(b_span.copy()),
                                    alpha=alpha, beta=beta)
            arrays[f"scale|{dt}|{alpha}|{beta}"] = s.numpy()
    # applying a scale
    for dt in ("bf16", "f16"):
        prev = nn.Linear(48, 64, bias=True).to(TORCH_DT[dt])
        mods = [nn.Linear(64, 40, bias=False).to(TORCH_DT[dt]), nn.Linear(64, 24, bias=False).to(TORCH_DT[dt])]
        for m in [prev] + mods:
            m.weight.data = (torch.randn(m.weight.shape, generator=g) * 0.07).to(TORCH_DT[dt])
        prev.bias.data = (torch.randn(64, generator=g) * 0.1).to(TORCH_DT[dt])
        scale = torch.exp(torch.randn(64, generator=g) * 0.7)
        arrays[f"apply|{dt}|scale"] = scale.numpy()
        arrays[f"apply|{dt}|prev_w"], arrays[f"apply|{dt}|prev_b"] = _np(prev.weight.data), _np(prev.bias.data)
        for i, m in enumerate(mods):
            arrays[f"apply|{dt}|w{i}"] = _np(m.weight.data)
        rs.smooth_linear_modules(prev, mods, scale=scale.clone())
        arrays[f"apply|{dt}|prev_w_out"], arrays[f"apply|{dt}|prev_b_out"] = _np(prev.weight.data), _np(prev.bias.data)
        for i, m in enumerate(mods):
            arrays[f"apply|{dt}|w{i}_out"] = _np(m.weight.data)
        # grouped-query layout: 4 heads x 16 channels on the consumer, 2 key/value heads on the producer
        prev2 = nn.Linear(48, 32, bias=False).to(TORCH_DT[dt])
        prev2.weight.data = (torch.randn(32, 48, generator=g) * 0.07).to(TORCH_DT[dt])
        mod2 = nn.Linear(64, 40, bias=False).to(TORCH_DT[dt])
        mod2.weight.data = (torch.randn(40, 64, generator=g) * 0.07).to(TORCH_DT[dt])
        s_half = torch.exp(torch.randn(2, 1, 16, generator=g) * 0.5).expand(2, 2, 16).reshape(-1).contiguous()
        arrays[f"apply_gqa|{dt}|scale"], arrays[f"apply_gqa|{dt}|prev_w"] = s_half.numpy(), _np(prev2.weight.data)
        arrays[f"apply_gqa|{dt}|w"] = _np(mod2.weight.data)
        rs.smooth_linear_modules(prev2, mod2, scale=s_half.clone(), num_heads=4, num_head_repeats=2)
        arrays[f"apply_gqa|{dt}|prev_w_out"], arrays[f"apply_gqa|{dt}|w_out"] = _np(prev2.weight.data), _np(mod2.weight.data)

    # a small search: two float32 linears fed by the same activation, sint4 weights (group 16), sint8 per-token inputs
    class Pair(nn.Module):
        def __init__(self, a, b):
            super().__init__()
            self.a, self.b = a, b

        def forward(self, x):
            return torch.cat([self.a(x), self.b(x)], dim=-1)

    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.quantizer import Quantizer as RefQuantizer
    from deepcompressor.quantizer.config.base import QuantizerConfig as RefQuantizerConfig

    for tag, strategy, alpha, beta in (("grid", SearchBasedCalibStrategy.GridSearch, 0.5, -2),
                                       ("manual", SearchBasedCalibStrategy.Manual, 0.3, 0.7)):
        a, b = nn.Linear(64, 40, bias=False), nn.Linear(64, 24, bias=False)
        a.weight.data = torch.randn(40, 64, generator=g) * 0.08
        b.weight.data = torch.randn(24, 64, generator=g) * 0.08
        gains = torch.exp(torch.randn(64, generator=g) * 1.2)
        xs = [torch.randn(2, 12, 64, generator=g) * gains for _ in range(3)]
        arrays[f"search|{tag}|wa"], arrays[f"search|{tag}|wb"] = a.weight.data.numpy().copy(), b.weight.data.numpy().copy()
        for i, x in enumerate(xs):
            arrays[f"search|{tag}|x{i}"] = x.numpy().copy()
        wq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint4"), zero_point=None,
                                             group_shapes=((1, 16),), scale_dtypes=(torch.float16,)), key="w")
        xq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None,
                                             group_shapes=((1, -1),), scale_dtypes=(torch.float16,)), channels_dim=-1,
                          key="x")
        cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=alpha, beta=beta,
                                num_grids=20, strategy=strategy, objective=SearchBasedCalibObjective.OutputsError,
                                granularity=SearchBasedCalibGranularity.Layer, degree=2)
        cache = TensorsCache(TensorCache(list(xs), channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3,
                                         num_samples=6, orig_device=torch.device("cpu")))
        calib = rs.SmoothLinearCalibrator(cfg, wq, xq)
        errors = []
        orig_tell = calib._tell

        def spy(error, _orig=orig_tell, _errors=errors):
            _errors.append(float(error[0]))
            return _orig(error)

        calib._tell = spy
        best = calib.calibrate(x_wgts=[a.weight, b.weight], x_acts=cache, x_mods=[a, b], eval_inputs=cache,
                               eval_module=Pair(a, b), eval_kwargs={})
        arrays[f"search|{tag}|best_scale"] = best.numpy().copy()
        arrays[f"search|{tag}|errors"] = np.array(errors, dtype=np.float64)
        meta[f"search|{tag}"] = dict(alpha=alpha, beta=beta, strategy=strategy.name, num_candidates=len(errors),
                                     best_index=int(np.argmin(errors)))
    np.savez_compressed(os.path.join(OUT, "smooth.npz"), **arrays)
    with open(os.path.join(OUT, "smooth.json"), "w") as f:
        json.dump(meta, f, indent=1)


SMOOTH_PRODUCTS_CASES = {
    # tag: objective, granularity, dtype, heads, repeats, quantize inputs, evaluation module is the linear itself
    "products_chgroup": ("ProductsError", "ChannelGroup", "f32", 1, 1, True, False),
    "products_group": ("ProductsError", "Group", "f32", 1, 1, False, False),  # the config turns Group into ChannelGroup
    "products_chgroup_f16": ("ProductsError", "ChannelGroup", "f16", 1, 1, False, False),
    "outputs_linear_chgroup": ("OutputsError", "ChannelGroup", "f32", 1, 1, True, True),
    "products_chgroup_gqa": ("ProductsError", "ChannelGroup", "f32", 4, 2, True, False),
    "products_chgroup_gqa_wide": ("ProductsError", "ChannelGroup", "f32", 8, 2, False, False),
}


SMOOTH_TENSOR_CASES = {
    # the same search scored on the weights alone (calib/search.py:690-696, 776-790)
    "tensor_chgroup": ("TensorError", "ChannelGroup", "f32", 1, 1, False, False),
    "tensor_chgroup_bf16": ("TensorError", "ChannelGroup", "bf16", 1, 1, False, False),
    "tensor_chgroup_f16": ("TensorError", "ChannelGroup", "f16", 1, 1, False, False),
    "tensor_layer": ("TensorError", "Layer", "f32", 1, 1, False, False),
    "tensor_chgroup_gqa": ("TensorError", "ChannelGroup", "f32", 4, 2, False, False),
}


def gen_smooth_tensor(ref) -> None:
    gen_smooth_products(ref, SMOOTH_TENSOR_CASES, "smooth_tensor", seed=4343)


def gen_smooth_products(ref, cases=None, out_name="smooth_products", seed=4242) -> None:
    """The reference's smoothing search scored on x-w products (calib/search.py:697-745, 791-816) with one winner per
    group of input channels (calib/smooth.py:655-714): every candidate's error tensors and the final scale."""
    import torch.nn as nn
    from deepcompressor.calib import smooth as rs
    from deepcompressor.calib.config import SearchBasedCalibGranularity, SearchBasedCalibObjective, SearchBasedCalibStrategy
    from deepcompressor.calib.config.smooth import SmoothCalibConfig, SmoothSpanMode
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.quantizer import Quantizer as RefQuantizer
    from deepcompressor.quantizer.config.base import QuantizerConfig as RefQuantizerConfig

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(seed)
    for tag, (objective, granularity, dt, heads, repeats, quant_x, linear_eval) in (cases or SMOOTH_PRODUCTS_CASES).items():
        dtype = TORCH_DT[dt]
        rows = (40,) if linear_eval else (40, 24)
        mods = [nn.Linear(64, r, bias=False).to(dtype) for r in rows]
        for m in mods:
            m.weight.data = (torch.randn(m.weight.shape, generator=g) * 0.08).to(dtype)
        gains = torch.exp(torch.randn(64, generator=g) * 1.2)
        xs = [(torch.randn(2, 12, 64, generator=g) * gains).to(dtype) for _ in range(3)]
        for i, m in enumerate(mods):
            arrays[f"{tag}|w{i}"] = _np(m.weight.data)
        for i, x in enumerate(xs):
            arrays[f"{tag}|x{i}"] = _np(x)
        wq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint4"), zero_point=None,
                                             group_shapes=((1, 16),), scale_dtypes=(torch.float16,)), key="w")
        xq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None,
                                             group_shapes=((1, -1),), scale_dtypes=(torch.float16,)), channels_dim=-1,
                          key="x") if quant_x else None
        cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=10,
                                strategy=SearchBasedCalibStrategy.GridSearch,
                                objective=SearchBasedCalibObjective[objective],
                                granularity=SearchBasedCalibGranularity[granularity], degree=2)
        cache = TensorsCache(TensorCache(list(xs), channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3,
                                         num_samples=6, orig_device=torch.device("cpu")))
        calib = rs.SmoothLinearCalibrator(cfg, wq, xq, num_heads=heads, num_head_repeats=repeats)
        errors, scales = [], []
        orig_tell = calib._tell

        def spy(error, _orig=orig_tell, _errors=errors, _scales=scales, _calib=calib):
            _errors.append(torch.stack([e.detach().clone().float().view(-1) for e in error]))
            _scales.append(_calib.candidate.detach().clone())
            return _orig(error)

        calib._tell = spy
        best = calib.calibrate(x_wgts=[m.weight for m in mods], x_acts=cache, x_mods=mods, eval_inputs=cache,
                               eval_module=mods[0] if linear_eval else None, eval_kwargs={})
        arrays[f"{tag}|errors"] = torch.stack(errors).numpy()  # [candidates, weights, groups or 1]
        arrays[f"{tag}|scales"] = torch.stack(scales).numpy()
        arrays[f"{tag}|best_scale"] = best.float().numpy().copy()
        meta[tag] = dict(objective=objective, granularity=granularity, dtype=dt, num_heads=heads, num_head_repeats=repeats,
                         quantize_inputs=quant_x, linear_eval=linear_eval, rows=list(rows),
                         num_candidates=len(errors))
    np.savez_compressed(os.path.join(OUT, f"{out_name}.npz"), **arrays)
    with open(os.path.join(OUT, f"{out_name}.json"), "w") as f:
        json.dump(meta, f, indent=1)


@torch.no_grad()
def gen_lowrank(ref) -> None:
    """Low-rank branches, joint quantization and the alternating calibration from the reference
    (deepcompressor/nn/patch/lowrank.py, quantizer/processor.py:215-310, calib/lowrank.py)."""
    import torch.nn as nn
    from deepcompressor.calib.config import SearchBasedCalibObjective
    from deepcompressor.calib.config.lowrank import QuantLowRankCalibConfig
    from deepcompressor.calib.lowrank import QuantLowRankCalibrator
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.nn.patch.lowrank import LowRankBranch
    from deepcompressor.quantizer import Quantizer as RefQuantizer
    from deepcompressor.quantizer.config.base import QuantizerConfig as RefQuantizerConfig
    from deepcompressor.quantizer.config.lowrank import QuantLowRankConfig

    arrays, meta = {}, {"dirnames": {}}
    g = torch.Generator().manual_seed(2024)

    def lowrank_weight(rows, cols, rank_signal=6):
        # a decaying spectrum so the leading singular subspace is well separated
        u = torch.linalg.qr(torch.randn(rows, rows, generator=g))[0][:, :rank_signal]
        v = torch.linalg.qr(torch.randn(cols, cols, generator=g))[0][:, :rank_signal]
        sv = torch.tensor([3.0, 2.2, 1.6, 1.1, 0.8, 0.55])[:rank_signal]
        return (u * sv) @ v.T * 0.2 + torch.randn(rows, cols, generator=g) * 0.02

    for dt in ("f32", "bf16"):
        w = lowrank_weight(48, 64).to(TORCH_DT[dt])
        lb = LowRankBranch(64, 48, rank=4, weight=w)
        arrays[f"branch|{dt}|w"] = _np(w)
        arrays[f"branch|{dt}|effective"] = _np(lb.get_effective_weight())
        arrays[f"branch|{dt}|singular"] = torch.linalg.svdvals(w.double()).numpy()
        x = torch.randn(5, 64, generator=g).to(TORCH_DT[dt])
        arrays[f"branch|{dt}|x"], arrays[f"branch|{dt}|y"] = _np(x), _np(lb(x))

    def wq_with(low_rank):
        return RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint4"), zero_point=None,
                                               group_shapes=((1, 16),), scale_dtypes=(torch.float16,)),
                            key="w", low_rank=low_rank)

    w1, w2 = lowrank_weight(40, 64), lowrank_weight(24, 64)
    arrays["joint|w1"], arrays["joint|w2"] = w1.numpy().copy(), w2.numpy().copy()
    for tag, kw in (("shared", dict(rank=4)), ("exclusive", dict(rank=4, exclusive=True)),
                    ("compensate", dict(rank=4, compensate=True)), ("exclusive_compensate", dict(rank=4, exclusive=True, compensate=True))):
        qts, lbs = wq_with(QuantLowRankConfig(**kw)).quantize_with_low_rank([w1.clone(), w2.clone()])
        for i, (qt, lb) in enumerate(zip(qts, lbs)):
            arrays[f"joint|{tag}|q{i}"] = _np(qt.data)
            arrays[f"joint|{tag}|l{i}"] = _np(lb.get_effective_weight())
    for kw in (dict(rank=8, num_iters=4, early_stop=True), dict(rank=16, exclusive=True), dict(rank=32, compensate=True),
               dict(rank=32, compensate=True, num_iters=3)):
        cfg = QuantLowRankCalibConfig(**kw)
        meta["dirnames"][json.dumps(kw, sort_keys=True)] = cfg.generate_dirnames(prefix="lr")

    # alternating calibration on one float32 linear
    lin = nn.Linear(64, 40, bias=False)
    lin.weight.data = lowrank_weight(40, 64)
    xs = [torch.randn(2, 10, 64, generator=g) for _ in range(3)]
    arrays["calib|w"] = lin.weight.data.numpy().copy()
    for i, x in enumerate(xs):
        arrays[f"calib|x{i}"] = x.numpy().copy()
    xq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None, group_shapes=((1, -1),),
                                         scale_dtypes=(torch.float16,)), channels_dim=-1, key="x")
    cfg = QuantLowRankCalibConfig(rank=4, num_iters=5, early_stop=False, objective=SearchBasedCalibObjective.OutputsError)
    cache = TensorsCache(TensorCache(list(xs), channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3,
                                     num_samples=6, orig_device=torch.device("cpu")))
    calib = QuantLowRankCalibrator(cfg, wq_with(None), xq)
    errors = []
    orig_tell = calib._tell

    def spy(error, _orig=orig_tell):
        errors.append(float(error[0] if isinstance(error, (list, tuple)) else error))
        return _orig(error)

    calib._tell = spy
    best = calib.calibrate(x_wgts=[lin.weight], x_mods=[lin], eval_inputs=cache, eval_module=lin, eval_kwargs={})
    arrays["calib|errors"] = np.array(errors, dtype=np.float64)
    arrays["calib|best_effective"] = _np(best.get_effective_weight())
    meta["calib"] = dict(num_iters=5, rank=4, best_index=int(np.argmin(errors)))
    np.savez_compressed(os.path.join(OUT, "lowrank.npz"), **arrays)
    with open(os.path.join(OUT, "lowrank.json"), "w") as f:
        json.dump(meta, f, indent=1)

@torch.no_grad()
def gen_branch(ref) -> None:
    """``branch.pt`` contents: the state dicts of the reference's LowRankBranch modules as
    app/diffusion/quant/weight.py:107-116 stores them, and what lines 117-144 of that file do with a stored entry
    (shared ``a``, ``b`` split over the modules that share their input, the effective weight subtracted from each
    module, the branch hooked behind it) -- run here with the reference's own LowRankBranch / AccumBranchHook."""
    import torch.nn as nn
    from deepcompressor.nn.patch.lowrank import LowRankBranch

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(77)
    rank = 4
    for tag, dt, outs in (("qkv_bf16", "bf16", (40, 24, 24)), ("single_f32", "f32", (48,)), ("pair_f16", "f16", (32, 16))):
        dtype = TORCH_DT[dt]
        mods = [nn.Linear(64, o, bias=False) for o in outs]
        for m in mods:
            m.weight.data = (torch.randn(m.weight.shape, generator=g) * 0.05
                             + torch.randn(m.weight.shape[0], 3, generator=g) @ torch.randn(3, 64, generator=g) * 0.1)
            m.to(dtype)
        x = torch.randn(6, 64, generator=g).to(dtype)
        for i, m in enumerate(mods):
            arrays[f"{tag}|w{i}"] = _np(m.weight.data).copy()  # (float32: .numpy() shares the tensor's memory)
        arrays[f"{tag}|x"] = _np(x).copy()
        joint = torch.cat([m.weight.data for m in mods], dim=0)
        state = LowRankBranch(64, joint.shape[0], rank=rank, weight=joint).state_dict()
        assert sorted(state) == ["a.weight", "b.weight"]
        for k, v in state.items():
            arrays[f"{tag}|state|{k}"] = _np(v).copy()
        # app/diffusion/quant/weight.py:117-144
        shared = LowRankBranch(in_features=64, out_features=joint.shape[0], rank=rank)
        shared.to(dtype=dtype)
        shared.load_state_dict(state)
        if len(mods) > 1:
            oc = 0
            for m in mods:
                branch = LowRankBranch(in_features=64, out_features=m.weight.shape[0], rank=rank)
                branch.a = shared.a
                branch.b.to(dtype=dtype)
                branch.b.weight.copy_(shared.b.weight[oc:oc + m.weight.shape[0]])
                oc += m.weight.shape[0]
                m.weight.data.sub_(branch.get_effective_weight().view(m.weight.data.shape))
                branch.as_hook().register(m)
        else:
            mods[0].weight.data.sub_(shared.get_effective_weight().view(mods[0].weight.data.shape))
            shared.as_hook().register(mods[0])
        for i, m in enumerate(mods):
            arrays[f"{tag}|after{i}"] = _np(m.weight.data).copy()
            arrays[f"{tag}|y{i}"] = _np(m(x)).copy()
        meta[tag] = dict(dtype=dt, outs=list(outs), rank=rank, in_features=64,
                         state_dtypes={k: str(v.dtype) for k, v in state.items()})
    np.savez_compressed(os.path.join(OUT, "branch.npz"), **arrays)
    with open(os.path.join(OUT, "branch.json"), "w") as f:
        json.dump(meta, f, indent=1)


RANGE_WIDE_EXTRA_CASES = {
    # groups longer than 512 values: per-channel rows of 2048
    "sint4_gchn_long": dict(dtype="sint4", zero=None, groups=((1, -1),), sdt=("f16",), shape=(16, 2048)),
}


def gen_range_wide(ref) -> None:
    """Layer / ChannelGroup granularity, long groups, multi-level scales and the per-step search of progressive
    quantizers (calib/range.py:302-474)."""
    import torch.nn as nn
    from deepcompressor.calib.config import (
        DynamicRangeCalibConfig,
        SearchBasedCalibGranularity,
        SearchBasedCalibObjective,
        SearchBasedCalibStrategy,
    )
    from deepcompressor.calib.range import DynamicRangeCalibrator, calibrate_dynamic_range
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.common import TensorType
    from deepcompressor.data.utils.reshape import LinearReshapeFn

    cases = dict(FAKE_QUANT_CASES, **RANGE_WIDE_EXTRA_CASES)
    arrays, meta = {}, {}
    records: list[dict] = []
    orig_tell = DynamicRangeCalibrator._tell

    def tell(self, error, _orig=orig_tell):
        if not records or records[-1]["owner"] is not self:
            records.append(dict(owner=self, errors=[]))
        records[-1]["errors"].append(_np(error[0]).copy())
        _orig(self, error)

    DynamicRangeCalibrator._tell = tell
    try:
        plan = [
            ("sint4_g128_f16s", "bf16", 2.4, 100, "Layer"), ("uint4_post_g128_f16s", "f16", 2, 80, "Layer"),
            ("sint4_g128_f16s", "f32", 2, 40, "Layer"), ("sint4_g128_f16s", "bf16", 2.4, 100, "ChannelGroup"),
            ("sint4_gchn_long", "bf16", 2.4, 40, "Group"), ("sint4_gchn_long", "f16", 2, 40, "Group"),
            ("sint4_gchn_g32_s8", "bf16", 2.4, 40, "Group"), ("nvfp4_g16", "bf16", 2, 40, "Group"),
            ("sint4_2x64", "bf16", 2, 40, "Group"),
            ("qoq_g128", "bf16", 2, 40, "Group"), ("qoq_g128", "f16", 2.4, 40, "Group"),
            ("qoq_g128_pre", "bf16", 2, 40, "Group"),
        ]
        for cname, dt, degree, grids, gran in plan:
            case = cases[cname]
            w = _mk_input(case["shape"], dt, seed=2000 + len(cname) + len(dt))
            q = _build_quantizer(ref, case)
            cfg = DynamicRangeCalibConfig(
                degree=degree, objective=SearchBasedCalibObjective.TensorError,
                strategy=SearchBasedCalibStrategy.GridSearch, granularity=SearchBasedCalibGranularity[gran],
                max_shrink=0.2, max_expand=1.0, num_grids=grids,
            )
            dummy = TensorsCache(TensorCache([torch.zeros(1, 8, w.shape[-1])], channels_dim=-1,
                                             reshape=LinearReshapeFn()))
            lin = nn.Linear(w.shape[1], w.shape[0], bias=False)
            lin.weight = nn.Parameter(w, requires_grad=False)
            tag = f"{cname}|{dt}|d{degree}|g{grids}|{gran}"
            records.clear()
            raised = None
            try:
                ranges = calibrate_dynamic_range(TensorType.Weights, cfg, True, q, [lin], dummy, eval_inputs=dummy)
            except (RuntimeError, ValueError) as exc:  # the reference's own failure is part of the record
                raised, ranges = str(exc).splitlines()[0][:200], ()
            arrays[f"{tag}|x"] = _np(w)
            arrays[f"{tag}|ratios"] = np.asarray([r for it in cfg.get_ratios() for r in it], dtype=np.float64)
            for step, (rng, rec) in enumerate(zip(ranges, records)):
                arrays[f"{tag}|s{step}|max"] = _np(rng.max)
                if rng.min is not None:
                    arrays[f"{tag}|s{step}|min"] = _np(rng.min)
                arrays[f"{tag}|s{step}|ratio"] = _np(rec["owner"].best_range.ratio)
                arrays[f"{tag}|s{step}|best_error"] = _np(rec["owner"].best_error)
                arrays[f"{tag}|s{step}|errors"] = np.stack(rec["errors"])
            if raised is not None and records:
                arrays[f"{tag}|s0|errors"] = np.stack(records[0]["errors"])  # candidates scored before the failure
            meta[tag] = dict(case=cname, in_dtype=dt, degree=degree, num_grids=grids, granularity=gran,
                             num_steps=len(ranges), reference_raised=raised,
                             codebook_via_restatement="fp" in case["dtype"] or any(
                                 isinstance(x, str) and x.startswith("sfp") for x in case["sdt"]))
    finally:
        DynamicRangeCalibrator._tell = orig_tell
    np.savez_compressed(os.path.join(OUT, "range_wide.npz"), **arrays)
    with open(os.path.join(OUT, "range_wide.json"), "w") as f:
        json.dump(dict(cases={k: {kk: (list(vv) if kk == "shape" else vv) for kk, vv in v.items()}
                              for k, v in RANGE_WIDE_EXTRA_CASES.items()}, entries=meta), f, indent=1)

def gen_range_products(ref) -> None:
    """Weight range search scored on products / outputs instead of the weight itself (calib/search.py:697-770,
    791-832): the AWQ-style ProductsError, OutputsError on a linear (which the reference turns into ProductsError
    below Layer granularity, calib/search.py:397-409) and OutputsError at Layer granularity."""
    import torch.nn as nn
    from deepcompressor.calib.config import (
        DynamicRangeCalibConfig,
        SearchBasedCalibGranularity,
        SearchBasedCalibObjective,
        SearchBasedCalibStrategy,
    )
    from deepcompressor.calib.range import DynamicRangeCalibrator, calibrate_dynamic_range
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.common import TensorType
    from deepcompressor.data.utils.reshape import LinearReshapeFn

    arrays, meta = {}, {}
    records: list[dict] = []
    orig_tell = DynamicRangeCalibrator._tell

    def tell(self, error, _orig=orig_tell):
        if not records or records[-1]["owner"] is not self:
            records.append(dict(owner=self, errors=[]))
        records[-1]["errors"].append(_np(error[0]).copy())
        _orig(self, error)

    DynamicRangeCalibrator._tell = tell
    try:
        plan = [
            ("sint4_g128_f16s", "f32", 2, 20, "ProductsError", "Group", 16),
            ("sint4_g128_f16s", "bf16", 2, 20, "ProductsError", "Group", 16),
            ("uint4_post_g128_f16s", "f16", 2, 20, "OutputsError", "Group", -1),
            ("uint4_post_g128_f16s", "f32", 2, 20, "OutputsError", "Group", -1),
            ("sint4_g128_f16s", "f32", 2, 20, "ProductsError", "ChannelGroup", -1),
            ("sint4_g128_f16s", "f32", 2, 20, "OutputsError", "Layer", -1),
            ("sint4_g128_f16s", "bf16", 2, 20, "OutputsError", "Layer", -1),
            ("uint4_pre_g128_f16s", "f32", 4, 10, "OutputsError", "Layer", -1),
        ]
        for cname, dt, degree, grids, objective, gran, ebatch in plan:
            case = FAKE_QUANT_CASES[cname]
            w = _mk_input(case["shape"], dt, seed=3000 + len(cname) + len(dt))
            g = torch.Generator().manual_seed(77 + len(cname))
            gains = torch.exp(torch.randn(w.shape[1], generator=g))
            xs = [(torch.randn(2, 16, w.shape[1], generator=g) * gains).to(TORCH_DT[dt]) for _ in range(2)]
            bias = (torch.randn(w.shape[0], generator=g) * 0.1).to(TORCH_DT[dt])
            q = _build_quantizer(ref, case)
            cfg = DynamicRangeCalibConfig(
                degree=degree, objective=SearchBasedCalibObjective[objective],
                strategy=SearchBasedCalibStrategy.GridSearch, granularity=SearchBasedCalibGranularity[gran],
                max_shrink=0.8, max_expand=1.0, num_grids=grids, element_batch_size=ebatch,
                element_size=-1, sample_batch_size=-1, sample_size=-1,
            )
            acts = TensorsCache(TensorCache([x.clone() for x in xs], channels_dim=-1, reshape=LinearReshapeFn(),
                                            num_cached=2, num_total=2, num_samples=4))
            lin = nn.Linear(w.shape[1], w.shape[0], bias=True).to(TORCH_DT[dt])
            lin.weight = nn.Parameter(w.clone(), requires_grad=False)
            lin.bias = nn.Parameter(bias, requires_grad=False)
            tag = f"{cname}|{dt}|d{degree}|g{grids}|{objective}|{gran}|e{ebatch}"
            records.clear()
            raised = None
            try:
                with torch.inference_mode():
                    ranges = calibrate_dynamic_range(TensorType.Weights, cfg, True, q, [lin], acts)
            except (RuntimeError, ValueError) as exc:
                raised, ranges = str(exc).splitlines()[0][:200], ()
            assert torch.equal(lin.weight.data, w), "the search must leave the weight as it found it"
            arrays[f"{tag}|w"] = _np(w)
            arrays[f"{tag}|bias"] = _np(bias)
            for i, x in enumerate(xs):
                arrays[f"{tag}|x{i}"] = _np(x)
            arrays[f"{tag}|ratios"] = np.asarray([r for it in cfg.get_ratios() for r in it], dtype=np.float64)
            for step, (rng, rec) in enumerate(zip(ranges, records)):
                arrays[f"{tag}|s{step}|max"] = _np(rng.max)
                if rng.min is not None:
                    arrays[f"{tag}|s{step}|min"] = _np(rng.min)
                arrays[f"{tag}|s{step}|ratio"] = _np(rec["owner"].best_range.ratio)
                arrays[f"{tag}|s{step}|best_error"] = _np(rec["owner"].best_error)
                arrays[f"{tag}|s{step}|errors"] = np.stack(rec["errors"])
            if raised is not None and records:
                arrays[f"{tag}|s0|errors"] = np.stack(records[0]["errors"])
            meta[tag] = dict(case=cname, in_dtype=dt, degree=degree, num_grids=grids, objective=objective,
                             granularity=gran, element_batch_size=ebatch, num_steps=len(ranges),
                             reference_raised=raised,
                             resolved_objective=records[0]["owner"].objective.name if records else None,
                             resolved_granularity=records[0]["owner"].granularity.name if records else None)
    finally:
        DynamicRangeCalibrator._tell = orig_tell
    np.savez_compressed(os.path.join(OUT, "range_products.npz"), **arrays)
    with open(os.path.join(OUT, "range_products.json"), "w") as f:
        json.dump(meta, f, indent=1)

class ToyScores(torch.nn.Module):
    """Query-key scores of a small grouped-query attention: the evaluation module of the query / key smoothing
    fixtures.  ``attn_q`` / ``attn_k`` are the (identity) modules whose outputs the calibrator quantizes."""

    def __init__(self, channels: int, heads: int, kv_heads: int, head_dim: int):
        super().__init__()
        nn = torch.nn
        self.heads, self.kv_heads, self.head_dim = heads, kv_heads, head_dim
        self.q_proj = nn.Linear(channels, heads * head_dim, bias=False)
        self.k_proj = nn.Linear(channels, kv_heads * head_dim, bias=False)
        self.attn_q, self.attn_k = nn.Identity(), nn.Identity()

    def forward(self, x):
        b, t, _ = x.shape
        q = self.attn_q(self.q_proj(x)).view(b, t, self.heads, self.head_dim).transpose(1, 2)
        k = self.attn_k(self.k_proj(x)).view(b, t, self.kv_heads, self.head_dim).transpose(1, 2)
        k = k.repeat_interleave(self.heads // self.kv_heads, dim=1)
        return q @ k.transpose(-1, -2)


@torch.no_grad()
def gen_smooth_attn(ref) -> None:
    """Query / key smoothing search (deepcompressor/calib/smooth.py:757-849, 1018-1109; calib/search.py:996-1078)."""
    from deepcompressor.calib import smooth as rs
    from deepcompressor.calib.config import SearchBasedCalibStrategy
    from deepcompressor.calib.config.smooth import SmoothCalibConfig, SmoothSpanMode
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.data.zero import ZeroPointDomain
    from deepcompressor.quantizer import Quantizer as RefQuantizer
    from deepcompressor.quantizer.config.base import QuantizerConfig as RefQuantizerConfig

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(4242)
    plan = [
        ("gqa_rope_k4", dict(heads=4, kv_heads=2, with_rope=True, quant_q=False, alpha=0.5, beta=0, spans=("AbsMax", "AbsMax"))),
        ("mha_norope_q8k4", dict(heads=4, kv_heads=4, with_rope=False, quant_q=True, alpha=0.5, beta=-2,
                                 spans=("AbsMax", "RootMeanSquare"))),
        ("gqa_manual", dict(heads=4, kv_heads=2, with_rope=True, quant_q=False, alpha=0.5, beta=0, spans=("AbsMax", "AbsMax"),
                            manual=True)),
    ]
    for tag, c in plan:
        channels, head_dim = 48, 16
        mod = ToyScores(channels, c["heads"], c["kv_heads"], head_dim)
        mod.q_proj.weight.data = torch.randn(mod.q_proj.weight.shape, generator=g) * 0.2
        mod.k_proj.weight.data = torch.randn(mod.k_proj.weight.shape, generator=g) * 0.2 * torch.exp(
            torch.randn(c["kv_heads"] * head_dim, 1, generator=g))
        xs = [torch.randn(2, 10, channels, generator=g) for _ in range(3)]
        queries = [mod.q_proj(x) for x in xs]
        keys = [mod.k_proj(x) for x in xs]
        cache = lambda data: TensorsCache(TensorCache([d.clone() for d in data], channels_dim=-1,  # noqa: E731
                                                      reshape=LinearReshapeFn(), num_cached=3, num_total=3, num_samples=6))
        kq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale,
                                             group_shapes=((1, head_dim),), scale_dtypes=(torch.float16,)),
                          channels_dim=-1, key="k")
        qq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None,
                                             group_shapes=((1, -1),), scale_dtypes=(torch.float16,)),
                          channels_dim=-1, key="q") if c["quant_q"] else None
        cfg = SmoothCalibConfig(spans=[tuple(SmoothSpanMode[m] for m in c["spans"])], alpha=c["alpha"], beta=c["beta"],
                                num_grids=8, degree=2,
                                strategy=SearchBasedCalibStrategy.Manual if c.get("manual") else SearchBasedCalibStrategy.GridSearch)
        errors, scales = [], []
        orig_tell = rs.SmoothCalibrator._tell

        def spy(self, error, _orig=orig_tell):
            errors.append(float(error[0]))
            scales.append(self.candidate.numpy().copy())
            return _orig(self, error)

        rs.SmoothCalibrator._tell = spy
        try:
            q0, k0 = mod.q_proj.weight.data.clone(), mod.k_proj.weight.data.clone()
            best = rs.smooth_attention(q_proj=mod.q_proj, k_proj=mod.k_proj, scale=None, config=cfg, query_quantizer=qq,
                                       key_quantizer=kq, queries=cache(queries), keys=cache(keys), attn_q=mod.attn_q,
                                       attn_k=mod.attn_k, eval_inputs=cache(xs), eval_module=mod, eval_kwargs={},
                                       num_heads=c["heads"], num_head_repeats=c["heads"] // c["kv_heads"],
                                       with_rope=c["with_rope"])
        finally:
            rs.SmoothCalibrator._tell = orig_tell
        arrays[f"{tag}|wq"], arrays[f"{tag}|wk"] = q0.numpy(), k0.numpy()
        arrays[f"{tag}|wq_out"], arrays[f"{tag}|wk_out"] = mod.q_proj.weight.data.numpy().copy(), mod.k_proj.weight.data.numpy().copy()
        for i, x in enumerate(xs):
            arrays[f"{tag}|x{i}"] = x.numpy()
        arrays[f"{tag}|best_scale"] = best.numpy().copy()
        arrays[f"{tag}|errors"] = np.array(errors, dtype=np.float64)
        arrays[f"{tag}|scales"] = np.stack(scales)
        meta[tag] = dict(c, channels=channels, head_dim=head_dim, num_candidates=len(errors),
                         best_index=int(np.argmin(errors)) if errors else -1)
    np.savez_compressed(os.path.join(OUT, "smooth_attn.npz"), **arrays)
    with open(os.path.join(OUT, "smooth_attn.json"), "w") as f:
        json.dump(meta, f, indent=1)

@torch.no_grad()
def gen_smooth_lowrank(ref) -> None:
    """Smoothing search whose candidates are scored with the low-rank branch in place (allow_low_rank,
    deepcompressor/calib/smooth.py:580-625), as the SVDQuant configs run it."""
    import torch.nn as nn
    from deepcompressor.calib import smooth as rs
    from deepcompressor.calib.config import SearchBasedCalibStrategy
    from deepcompressor.calib.config.smooth import SmoothCalibConfig, SmoothSpanMode
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.quantizer import Quantizer as RefQuantizer
    from deepcompressor.quantizer.config.base import QuantizerConfig as RefQuantizerConfig
    from deepcompressor.quantizer.config.lowrank import QuantLowRankConfig

    class Pair(nn.Module):
        def __init__(self, a, b):
            super().__init__()
            self.a, self.b = a, b

        def forward(self, x):
            return torch.cat([self.a(x), self.b(x)], dim=-1)

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(777)
    for tag, lr_kw, splits in (("shared", dict(rank=4), None), ("exclusive", dict(rank=4, exclusive=True), None),
                               ("split", dict(rank=4), [1])):
        a, b = nn.Linear(64, 40, bias=False), nn.Linear(64, 24, bias=False)
        a.weight.data = torch.randn(40, 64, generator=g) * 0.08
        b.weight.data = torch.randn(24, 64, generator=g) * 0.08
        gains = torch.exp(torch.randn(64, generator=g) * 1.2)
        xs = [torch.randn(2, 12, 64, generator=g) * gains for _ in range(3)]
        arrays[f"{tag}|wa"], arrays[f"{tag}|wb"] = a.weight.data.numpy().copy(), b.weight.data.numpy().copy()
        for i, x in enumerate(xs):
            arrays[f"{tag}|x{i}"] = x.numpy().copy()
        wq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint4"), zero_point=None,
                                             group_shapes=((1, 16),), scale_dtypes=(torch.float16,)), key="w",
                          low_rank=QuantLowRankConfig(**lr_kw))
        xq = RefQuantizer(RefQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None,
                                             group_shapes=((1, -1),), scale_dtypes=(torch.float16,)), channels_dim=-1,
                          key="x")
        cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=6,
                                strategy=SearchBasedCalibStrategy.GridSearch, degree=2, allow_low_rank=True)
        cache = TensorsCache(TensorCache(list(xs), channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3,
                                         num_samples=6, orig_device=torch.device("cpu")))
        calib = rs.SmoothLinearCalibrator(cfg, wq, xq)
        errors = []
        orig_tell = calib._tell

        def spy(error, _orig=orig_tell, _errors=errors):
            _errors.append(float(error[0]))
            return _orig(error)

        calib._tell = spy
        best = calib.calibrate(x_wgts=[a.weight, b.weight], x_acts=cache, x_mods=[a, b], eval_inputs=cache,
                               eval_module=Pair(a, b), eval_kwargs={}, splits=splits)
        arrays[f"{tag}|best_scale"] = best.numpy().copy()
        arrays[f"{tag}|errors"] = np.array(errors, dtype=np.float64)
        meta[tag] = dict(low_rank=lr_kw, splits=splits, num_candidates=len(errors), best_index=int(np.argmin(errors)))
    np.savez_compressed(os.path.join(OUT, "smooth_lowrank.npz"), **arrays)
    with open(os.path.join(OUT, "smooth_lowrank.json"), "w") as f:
        json.dump(meta, f, indent=1)


class ToyGated(torch.nn.Module):
    """up(x) * sigmoid(gate(x)) -> down: the evaluation module of the activation range fixtures (``up`` and ``gate``
    share their input; ``mid`` is the identity whose output feeds ``down``)."""

    def __init__(self, channels: int, hidden: int):
        super().__init__()
        nn = torch.nn
        self.up = nn.Linear(channels, hidden, bias=False)
        self.gate = nn.Linear(channels, hidden, bias=False)
        self.mid = nn.Identity()
        self.down = nn.Linear(hidden, channels, bias=False)

    def forward(self, x):
        return self.down(self.mid(self.up(x) * torch.sigmoid(self.gate(x))))


RANGE_ACTS_CASES = {
    "sint8_tnsr": dict(dtype="sint8", zero=None, groups=((-1, -1),), sdt=("f16",)),
    "sint8_tnsr_defs": dict(dtype="sint8", zero=None, groups=((-1, -1),), sdt=(None,)),
    "uint4_post_g32": dict(dtype="uint4", zero="PostScale", groups=((1, 32),), sdt=("f16",)),
    "sint4_g32": dict(dtype="sint4", zero=None, groups=((1, 32),), sdt=("f16",)),
    "uint8_pre_tnsr": dict(dtype="uint8", zero="PreScale", groups=((-1, -1),), sdt=("f16",)),
    "sint8_token": dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",)),
}


RANGE_ACTS_PLAN = [
    # (quantizer case, dtype, tensor type, objective, granularity, strategy, static, allow_scale, degree, grids)
    ("sint8_tnsr", "bf16", "Inputs", "TensorError", "Group", "GridSearch", True, False, 2, 20),
    ("uint4_post_g32", "f16", "Inputs", "TensorError", "Group", "GridSearch", True, False, 2.4, 20),
    ("sint4_g32", "bf16", "Inputs", "TensorError", "ChannelGroup", "GridSearch", True, False, 2, 20),
    ("uint4_post_g32", "bf16", "Inputs", "TensorError", "Layer", "GridSearch", True, False, 2, 20),
    ("sint8_tnsr", "f32", "Inputs", "OutputsError", "Layer", "GridSearch", True, False, 2, 20),
    ("uint8_pre_tnsr", "bf16", "Inputs", "OutputsError", "Layer", "GridSearch", True, False, 2, 12),
    ("sint8_tnsr_defs", "f16", "Outputs", "OutputsError", "Layer", "Manual", True, False, 2, 20),
    ("sint8_tnsr", "f32", "Outputs", "OutputsError", "Layer", "GridSearch", True, False, 2, 20),
    ("uint4_post_g32", "f32", "Outputs", "TensorError", "Group", "GridSearch", True, False, 2, 12),
    ("sint8_token", "f32", "Inputs", "OutputsError", "Layer", "GridSearch", False, False, 2, 12),
    ("sint8_token", "f32", "Inputs", "OutputsError", "Layer", "GridSearch", False, True, 2, 12),
    ("sint4_g32", "bf16", "Inputs", "TensorError", "Layer", "GridSearch", False, True, 2, 12),
]
RANGE_ACTS_PRODUCTS_PLAN = [
    # x-w products of the linears that take the input (calib/search.py:871-901, 947-973)
    ("sint8_tnsr", "f32", "Inputs", "ProductsError", "ChannelGroup", "GridSearch", True, False, 2, 12),
    ("sint4_g32", "f32", "Inputs", "ProductsError", "ChannelGroup", "GridSearch", True, False, 2, 12),
    ("uint4_post_g32", "f16", "Inputs", "ProductsError", "ChannelGroup", "GridSearch", True, False, 2, 12),
    ("uint4_post_g32", "f32", "Inputs", "ProductsError", "Group", "GridSearch", True, False, 2, 12),
    ("sint8_token", "f32", "Inputs", "ProductsError", "ChannelGroup", "GridSearch", False, True, 2, 12),
    ("sint4_g32", "f32", "Inputs", "ProductsError", "Group", "GridSearch", False, False, 2, 12),
]


def gen_range_acts(ref) -> None:
    _gen_range_acts(ref, RANGE_ACTS_PLAN, "range_acts", 0)


def gen_range_acts_products(ref) -> None:
    _gen_range_acts(ref, RANGE_ACTS_PRODUCTS_PLAN, "range_acts_products", 100)


@torch.no_grad()
def _gen_range_acts(ref, plan, fname, first_index) -> None:
    """Range search of activations (calib/range.py:101-150, calib/search.py:839-1078): static ranges of a module's
    inputs scored on the tensor itself or on an evaluation module's outputs, static ranges of a module's outputs,
    and the clamp / ratio search of dynamically quantized inputs."""
    from deepcompressor.calib.config import (
        DynamicRangeCalibConfig,
        SearchBasedCalibGranularity,
        SearchBasedCalibObjective,
        SearchBasedCalibStrategy,
    )
    from deepcompressor.calib.range import DynamicRangeCalibrator, calibrate_dynamic_range
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.common import TensorType
    from deepcompressor.data.utils.reshape import LinearReshapeFn

    arrays, meta = {}, {}
    records: list[dict] = []
    orig_tell = DynamicRangeCalibrator._tell

    def tell(self, error, _orig=orig_tell):
        if not records or records[-1]["owner"] is not self:
            records.append(dict(owner=self, errors=[]))
        records[-1]["errors"].append(_np(error[0]).copy())
        _orig(self, error)

    DynamicRangeCalibrator._tell = tell
    try:
        channels, hidden = 64, 96
        for idx, (cname, dt, ttype, objective, gran, strategy, static, allow_scale, degree, grids) in enumerate(plan, first_index):
            case = RANGE_ACTS_CASES[cname]
            g = torch.Generator().manual_seed(5000 + idx)
            tdt = TORCH_DT[dt]
            mod = ToyGated(channels, hidden).to(tdt)
            for lin in (mod.up, mod.gate, mod.down):
                lin.weight.data = (torch.randn(lin.weight.shape, generator=g) * 0.15).to(tdt)
            gains = torch.exp(torch.randn(channels, generator=g) * 0.8)
            xs = [(torch.randn(2, 12, channels, generator=g) * gains).to(tdt) for _ in range(3)]
            xs[1][0, 3, 5] = xs[1].abs().max() * 3  # an outlier, so that clipping pays
            cache = lambda data: TensorsCache(TensorCache([d.clone() for d in data], channels_dim=-1,  # noqa: E731
                                                          reshape=LinearReshapeFn(), num_cached=3, num_total=3,
                                                          num_samples=6))
            q = _build_quantizer(ref, case)
            q.channels_dim = -1
            cfg = DynamicRangeCalibConfig(
                degree=degree, objective=SearchBasedCalibObjective[objective],
                strategy=SearchBasedCalibStrategy[strategy], granularity=SearchBasedCalibGranularity[gran],
                max_shrink=0.4, max_expand=1.0, num_grids=grids, allow_scale=allow_scale,
                element_batch_size=-1, element_size=-1, sample_batch_size=-1, sample_size=-1,
            )
            if ttype == "Inputs":
                modules, acts = [mod.up, mod.gate], cache(xs)
            else:
                modules = [mod.mid]
                acts = cache([mod.up(x) * torch.sigmoid(mod.gate(x)) for x in xs])
            tag = f"{idx}|{cname}|{dt}|{ttype}|{objective}|{gran}|{strategy}|static{int(static)}|scale{int(allow_scale)}"
            records.clear()
            raised = None
            try:
                with torch.inference_mode():
                    ranges = calibrate_dynamic_range(TensorType[ttype], cfg, static, q, modules, acts,
                                                     eval_inputs=cache(xs), eval_module=mod, eval_kwargs={})
            except (RuntimeError, ValueError, AssertionError) as exc:
                raised, ranges = (str(exc).splitlines() or [type(exc).__name__])[0][:200], ()
            for name in ("up", "gate", "down"):
                arrays[f"{tag}|w_{name}"] = _np(getattr(mod, name).weight.data)
            for i, x in enumerate(xs):
                arrays[f"{tag}|x{i}"] = _np(x)
            arrays[f"{tag}|ratios"] = np.asarray([r for it in cfg.get_ratios() for r in it], dtype=np.float64)
            for step, (rng, rec) in enumerate(zip(ranges or (), records)):
                if rng.max is not None:
                    arrays[f"{tag}|s{step}|max"] = _np(rng.max)
                if isinstance(rng.min, torch.Tensor):
                    arrays[f"{tag}|s{step}|min"] = _np(rng.min)
                if isinstance(rng.ratio, torch.Tensor):
                    arrays[f"{tag}|s{step}|out_ratio"] = _np(rng.ratio)
                arrays[f"{tag}|s{step}|ratio"] = _np(rec["owner"].best_range.ratio)
                arrays[f"{tag}|s{step}|best_error"] = _np(rec["owner"].best_error)
                arrays[f"{tag}|s{step}|errors"] = np.stack(rec["errors"])
            if raised is not None and records:
                arrays[f"{tag}|s0|errors"] = np.stack(records[0]["errors"])
            meta[tag] = dict(case=cname, in_dtype=dt, tensor_type=ttype, objective=objective, granularity=gran,
                             strategy=strategy, static=static, allow_scale=allow_scale, degree=degree, num_grids=grids,
                             channels=channels, hidden=hidden, num_steps=len(ranges or ()), reference_raised=raised,
                             out_ratio_is_float=bool(ranges) and isinstance(ranges[0].ratio, float),
                             resolved_objective=records[0]["owner"].objective.name if records else None,
                             resolved_granularity=records[0]["owner"].granularity.name if records else None)
            print(" ", tag, "->", raised or "ok")
    finally:
        DynamicRangeCalibrator._tell = orig_tell
    np.savez_compressed(os.path.join(OUT, f"{fname}.npz"), **arrays)
    with open(os.path.join(OUT, f"{fname}.json"), "w") as f:
        json.dump(dict(cases=RANGE_ACTS_CASES, entries=meta), f, indent=1)


@torch.no_grad()
def gen_reorder(ref) -> None:
    """Channel reordering (deepcompressor/calib/reorder.py:61-540): every channel metric, both index modes, heads
    with and without repeats, and the calibrator (Manual as QoQ runs it, GridSearch over every metric / index mode)
    on a linear and on the gated toy module."""
    import torch.nn as nn
    from deepcompressor.calib import reorder as rr
    from deepcompressor.calib.config import SearchBasedCalibStrategy
    from deepcompressor.calib.config.reorder import ChannelOrderCalibConfig
    from deepcompressor.calib.metric import ChannelMetric
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.utils.reshape import LinearReshapeFn

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(9191)
    channels = 64
    gains = torch.exp(torch.randn(channels, generator=g))
    xs = [torch.randn(2, 10, channels, generator=g) * gains for _ in range(3)]
    ws = [torch.randn(48, channels, generator=g) * 0.1 * torch.exp(0.5 * torch.randn(channels, generator=g)),
          torch.randn(24, channels, generator=g) * 0.1]
    cache = lambda data: TensorsCache(TensorCache([d.clone() for d in data], channels_dim=-1,  # noqa: E731
                                                  reshape=LinearReshapeFn(), num_cached=3, num_total=3, num_samples=6))
    for i, x in enumerate(xs):
        arrays[f"x{i}"] = x.numpy().copy()
    for i, w in enumerate(ws):
        arrays[f"w{i}"] = w.numpy().copy()
    # --- metrics and the orders they give
    for mode in ChannelOrderCalibConfig.ChannelMetric:
        for heads in (1, 4):
            arrays[f"metric|{mode.name}|h{heads}"] = rr.get_channel_metric(cache(xs), ws, mode, channels,
                                                                           num_heads=heads).numpy().copy()
    arrays["metric|abs_normalize_mean|h4"] = ChannelMetric.abs_normalize_mean(
        [x.view(-1, channels) for x in xs], channels, [-1, channels // 4]).numpy().copy()
    metric = torch.from_numpy(arrays["metric|InputsAbsMax|h1"])
    index_cases = {}
    for imode in ChannelOrderCalibConfig.ChannelIndex:
        for heads, repeats in ((1, 1), (4, 1), (4, 2)):
            for mmode in ("InputsAbsMax", "InputsAbsMean"):
                key = f"index|{mmode}|{imode.name}|h{heads}|r{repeats}"
                arrays[key] = rr.init_channel_index_from_metric(
                    torch.from_numpy(arrays[f"metric|{mmode}|h1"]), ChannelOrderCalibConfig.ChannelMetric[mmode], imode,
                    group_size=16, num_heads=heads, num_head_repeats=repeats).numpy().copy()
                index_cases[key] = dict(metric=mmode, index_mode=imode.name, heads=heads, repeats=repeats, group_size=16)
    meta["index_cases"] = index_cases
    # --- the calibrator
    def quantizers(small: bool):
        # with heads, channels only move inside their head (16 channels): groups of 8 so that the order matters
        case_w = dict(dtype="uint4", zero="PostScale", groups=((1, 8 if small else 16),), sdt=("f16",))
        case_x = dict(dtype="sint8", zero=None, groups=((1, 8 if small else 32),), sdt=("f16",))
        wq, xq = _build_quantizer(ref, case_w), _build_quantizer(ref, case_x)
        xq.channels_dim = -1
        return wq, xq, case_w, case_x

    runs = {}
    for tag, strategy, module_kind, heads, repeats in (("manual_linear", "Manual", "linear", 1, 1),
                                                       ("search_linear", "GridSearch", "linear", 1, 1),
                                                       ("search_linear_heads", "GridSearch", "linear", 4, 2),
                                                       ("search_gated", "GridSearch", "gated", 1, 1)):
        wq, xq, case_w, case_x = quantizers(heads > 1)
        cfg = ChannelOrderCalibConfig(strategy=SearchBasedCalibStrategy[strategy],
                                      channel_metric=ChannelOrderCalibConfig.ChannelMetric.InputsAbsMax,
                                      channel_index=ChannelOrderCalibConfig.ChannelIndex.Sequential)
        calib = rr.ChannelOrderCalibrator(cfg, wq, xq, num_heads=heads, num_head_repeats=repeats)
        stats = []
        orig_tell = calib._tell

        def spy(errors, _orig=orig_tell, _calib=calib, _stats=stats):
            _orig(errors)
            _stats.append(float(errors[0][0]))

        calib._tell = spy
        if module_kind == "linear":
            lin = nn.Linear(channels, 48, bias=False)
            lin.weight.data = ws[0].clone()
            best = calib.calibrate(x_wgts=[lin.weight], x_acts=cache(xs), x_mods=[lin], eval_inputs=cache(xs),
                                   eval_module=lin, reorder_wgts=[(lin.weight, 1)], reorder_ipt_mods=[(lin, -1, None)],
                                   reorder_opt_mods=[])
            assert torch.equal(lin.weight.data, ws[0])
        else:
            mod = ToyGated(channels, 48)
            mod.up.weight.data, mod.gate.weight.data = ws[0].clone(), (ws[0] * 0.5 + 0.01).clone()
            mod.down.weight.data = torch.randn(channels, 48, generator=g) * 0.1
            arrays[f"{tag}|w_down"] = mod.down.weight.data.numpy().copy()
            best = calib.calibrate(x_wgts=[mod.up.weight, mod.gate.weight], x_acts=cache(xs), x_mods=[mod.up, mod.gate],
                                   eval_inputs=cache(xs), eval_module=mod, eval_kwargs={},
                                   reorder_wgts=[(mod.up.weight, 1), (mod.gate.weight, 1)],
                                   reorder_ipt_mods=[(mod, -1, None)], reorder_opt_mods=[])
        arrays[f"{tag}|errors"] = np.asarray(stats, dtype=np.float64)
        arrays[f"{tag}|best"] = (best.numpy().copy() if best is not None else np.zeros(0, dtype=np.int64))
        for i, idx in enumerate(calib.channel_indexes):
            arrays[f"{tag}|cand{i}"] = idx.numpy().copy() if idx is not None else np.zeros(0, dtype=np.int64)
        runs[tag] = dict(strategy=strategy, module=module_kind, heads=heads, repeats=repeats, case_w=case_w,
                         case_x=case_x, group_size=min(case_w["groups"][0][1], case_x["groups"][0][1]), best_candidate_id=int(calib.best_candidate_id), best_is_identity=best is None,
                         num_candidates=len(stats), best_error=[float(v) for v in calib.best_error])
        print(" ", tag, "best", calib.best_candidate_id, "of", len(stats))
    meta["runs"] = runs
    meta["channels"] = channels
    np.savez_compressed(os.path.join(OUT, "reorder.npz"), **arrays)
    with open(os.path.join(OUT, "reorder.json"), "w") as f:
        json.dump(meta, f, indent=1)


@torch.no_grad()
def gen_llm_stage(ref) -> None:
    """The per-layer weight stage as the reference's LLM app runs it (app/llm/quant/weight.py:57-103 with
    LlmWeightQuantizer): calibrate -> Quantizer.state_dict -> reload -> quantize -> scale state dict.  RTN kernel (the
    GPTQ kernel is covered by its own fixtures), GPTQ-R style range search and the QoQ progressive quantizer."""
    import torch.nn as nn
    from deepcompressor.app.llm.quant.quantizer.config import LlmWeightQuantizerConfig
    from deepcompressor.app.llm.quant.quantizer.quantizer import LlmWeightQuantizer
    from deepcompressor.calib.config import (
        SearchBasedCalibGranularity,
        SearchBasedCalibObjective,
        SearchBasedCalibStrategy,
        SkipBasedDynamicRangeCalibConfig,
    )
    from deepcompressor.data.cache import TensorCache, TensorsCache
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.data.zero import ZeroPointDomain

    arrays, meta = {}, {}
    plans = {
        "gptq_r_rtn": dict(dtype="uint4", zero="PostScale", groups=((1, 128),), sdt=("f16",), search=True),
        "qoq_g128": dict(dtype="uint4", zero="PostScale", groups=((1, -1), (1, 128)), sdt=("f16", "sint8"),
                         inter=("sint8",), levels=(0,), search=False),
        "qoq_gchn_pre": dict(dtype="sint4", zero="PreScale", groups=((1, -1),), sdt=("f16",), search=True, skip="k_proj"),
    }
    g = torch.Generator().manual_seed(31337)
    names = [("q_proj", "layers.0.attn.q_proj", (64, 256)), ("k_proj", "layers.0.attn.k_proj", (32, 256)),
             ("down_proj", "layers.0.mlp.down_proj", (48, 384))]
    for tag, c in plans.items():
        def sd(x):
            return None if x is None else (TORCH_DT[x] if x in TORCH_DT else QuantDataType.from_str(x))

        calib = SkipBasedDynamicRangeCalibConfig(
            degree=2.4, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
            granularity=SearchBasedCalibGranularity.Group, max_shrink=0.2, max_expand=1.0, num_grids=20,
            skips=[c["skip"]] if c.get("skip") else []) if c["search"] else None
        kw = dict(dtype=QuantDataType.from_str(c["dtype"]), zero_point=ZeroPointDomain[c["zero"]],
                  group_shapes=c["groups"], scale_dtypes=tuple(sd(x) for x in c["sdt"]), calib_range=calib)
        if "inter" in c:
            kw.update(intermediate_dtypes=tuple(QuantDataType.from_str(x) for x in c["inter"]),
                      intermediate_levels=c["levels"])
        config = LlmWeightQuantizerConfig(**kw)
        meta[tag] = dict(case={k: v for k, v in c.items()}, calib_dirname=config.generate_calib_dirname(),
                         dirnames=config.generate_dirnames(prefix="w"), modules=[])
        scale_state_dict = {}
        for key, name, shape in names:
            w = _mk_input(shape, "f16", seed=int(torch.randint(0, 10 ** 6, (1,), generator=g)))
            lin = nn.Linear(shape[1], shape[0], bias=False).to(torch.float16)
            lin.weight = nn.Parameter(w.clone(), requires_grad=False)
            dummy = TensorsCache(TensorCache([torch.zeros(1, 8, shape[1], dtype=torch.float16)], channels_dim=-1,
                                             reshape=LinearReshapeFn()))
            q = LlmWeightQuantizer(config, develop_dtype=torch.float32, key=key)
            q.calibrate_dynamic_range(module=lin, inputs=dummy, eval_inputs=dummy, eval_module=lin)
            state = q.state_dict()
            q2 = LlmWeightQuantizer(config, develop_dtype=torch.float32, key=key)
            q2.load_state_dict(state, device="cpu")
            result = q2.quantize(lin.weight.data, inputs=None, return_with_dequant=True, return_with_quant=True)
            arrays[f"{tag}|{name}|w"] = _np(w)
            arrays[f"{tag}|{name}|dequant"] = _np(result.data)
            entries = result.scale.state_dict(f"{name}.weight.scale")
            zero_name = "scaled_zero" if config.zero_domain is ZeroPointDomain.PostScale else "zero"
            entries[f"{name}.weight.{zero_name}"] = result.zero
            for k, v in entries.items():
                arrays[f"{tag}|sd|{k}"] = _np(v)
            dyn = state["dynamic_range"]
            dyn = (dyn,) if isinstance(dyn, dict) else (dyn or ())
            for step, d in enumerate(dyn):
                for part in ("min", "max"):
                    if d is not None and isinstance(d.get(part), torch.Tensor):
                        arrays[f"{tag}|{name}|range{step}|{part}"] = _np(d[part])
            meta[tag]["modules"].append(dict(key=key, name=name, shape=list(shape), sd_keys=sorted(entries),
                                             state_keys=sorted(state), num_ranges=len(dyn),
                                             range_is_none=state["dynamic_range"] is None))
    np.savez_compressed(os.path.join(OUT, "llm_stage.npz"), **arrays)
    with open(os.path.join(OUT, "llm_stage.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_round_delta(ref) -> None:
    """Quantization with a rounding offset instead of round-to-nearest (quantizer/impl/simple.py:44-51,
    kernel/rtn.py:103-117): codes = clamp(floor(t) + round_delta)."""
    arrays, meta = {}, {}
    for cname, dt in (("sint4_g128_f16s", "bf16"), ("uint4_post_g128_f16s", "f16"), ("uint4_pre_g128_f16s", "f32"),
                      ("qoq_g128", "f16"), ("sint8_tnsr_f16s", "bf16")):
        case = FAKE_QUANT_CASES[cname]
        x = _mk_input(case["shape"], dt, seed=77 + len(cname))
        g = torch.Generator().manual_seed(len(cname))
        delta = (torch.rand(*case["shape"], generator=g) < 0.5).to(torch.float32)  # hard 0 / 1 decisions
        soft = torch.rand(*case["shape"], generator=g)  # and a soft (fractional) offset
        for kind, d in (("hard", delta), ("soft", soft)):
            q = _build_quantizer(ref, case)
            # channels_dim=1: without it the reference prepends a dimension to round_delta (impl/base.py:107-109) and
            # then takes it for a single shared row (:230-234)
            codes = q.quantize(x, return_with_dequant=False, return_with_quant=True, round_delta=d, channels_dim=1).qdata
            q = _build_quantizer(ref, case)
            deq = q.quantize(x, return_with_dequant=True, return_with_quant=False, round_delta=d, channels_dim=1).data
            tag = f"{cname}|{dt}|{kind}"
            arrays[f"{tag}|x"], arrays[f"{tag}|delta"] = _np(x), _np(d)
            arrays[f"{tag}|codes"], arrays[f"{tag}|dequant"] = _np(codes), _np(deq)
            meta[tag] = dict(case=cname, in_dtype=dt, kind=kind)
    np.savez_compressed(os.path.join(OUT, "round_delta.npz"), **arrays)
    with open(os.path.join(OUT, "round_delta.json"), "w") as f:
        json.dump(meta, f, indent=1)


EXP_SCALE_CASES = {
    # power-of-two (exponent-only) scale levels (quantizer/impl/scale.py:124-140, 204-221)
    "sint4_g32_e8": dict(dtype="sint4", zero=None, groups=((1, 32),), sdt=("uexp8_all",), shape=(32, 256)),
    "sint4_gchn_f16_g32_e4": dict(dtype="sint4", zero=None, groups=((1, -1), (1, 32)), sdt=("f16", "uexp4_all"),
                                  shape=(32, 256)),
    "uint4_post_gchn_f16_g32_e4": dict(dtype="uint4", zero="PostScale", groups=((1, -1), (1, 32)),
                                       sdt=("f16", "uexp4_all"), shape=(32, 256)),
    "sfp4_gchn_e8_g16_e4": dict(dtype="sfp4_e2m1_all", zero=None, groups=((1, -1), (1, 16)),
                                sdt=("uexp8_all", "uexp4_all"), shape=(16, 128)),
    "uint4_pre_g64_e8": dict(dtype="uint4", zero="PreScale", groups=((1, 64),), sdt=("uexp8_all",), shape=(16, 256)),
}


def gen_exp_scale(ref) -> None:
    arrays, meta = {}, {}
    for cname, case in EXP_SCALE_CASES.items():
        for dt in ("bf16", "f16"):
            x = _mk_input(case["shape"], dt, seed=500 + len(cname)) * (1.0 if "e8" in cname else 8.0)
            x = x.to(TORCH_DT[dt])
            q = _build_quantizer(ref, case)
            r = q.quantize(x, return_with_dequant=False, return_with_quant=True)
            q = _build_quantizer(ref, case)
            deq = q.quantize(x, return_with_dequant=True, return_with_quant=False).data
            tag = f"{cname}|{dt}"
            arrays[f"{tag}|x"], arrays[f"{tag}|codes"], arrays[f"{tag}|dequant"] = _np(x), _np(r.qdata), _np(deq)
            arrays[f"{tag}|scale"] = _np(r.scale.data)
            for k, v in r.scale.state_dict("s").items():
                arrays[f"{tag}|leaf|{k}"] = _np(v)
            arrays[f"{tag}|zero"] = _np(r.zero)
            meta[tag] = dict(case=cname, in_dtype=dt, codebook_via_restatement="fp" in case["dtype"])
    np.savez_compressed(os.path.join(OUT, "exp_scale.npz"), **arrays)
    with open(os.path.join(OUT, "exp_scale.json"), "w") as f:
        json.dump(dict(cases={k: {kk: (list(vv) if kk == "shape" else vv) for kk, vv in v.items()}
                              for k, v in EXP_SCALE_CASES.items()}, entries=meta), f, indent=1)


@torch.no_grad()
def gen_qserve(ref) -> None:
    """The reference's QServe converter (backend/qserve/utils.py:10-210, convert.py:83-146) run on weight-stage
    outputs of the reference: per-group W4 (QoQ progressive, int8 zero and PreScale int4 zero), per-channel W4, W8."""
    from deepcompressor.backend.qserve.convert import convert_to_qserve_state_dict
    from deepcompressor.backend.qserve.utils import QServePacker

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(2718)
    codes = torch.randint(0, 16, (96, 160), generator=g, dtype=torch.uint8)
    arrays["pack|codes"], arrays["pack|packed"] = codes.numpy().copy(), QServePacker().pack_weight(codes).numpy().copy()
    plans = {
        "w4_g128_post": (dict(dtype="uint4", zero="PostScale", groups=((1, -1), (1, 128)), sdt=("f16", "sint8"),
                              inter=("sint8",), levels=(0,)), 4),
        "w4_g128_pre": (dict(dtype="uint4", zero="PreScale", groups=((1, -1), (1, 128)), sdt=("f16", "sint8"),
                             inter=("sint8",), levels=(0,)), 4),
        "w4_gchn_post": (dict(dtype="uint4", zero="PostScale", groups=((1, -1),), sdt=("f16",)), 4),
        "w4_gchn_pre": (dict(dtype="sint4", zero="PreScale", groups=((1, -1),), sdt=("f16",)), 4),
        "w8_gchn": (dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",)), 8),
    }
    for tag, (case, bits) in plans.items():
        state, scales = {}, {}
        for name, shape in (("layers.0.q_proj", (64, 256)), ("layers.0.down_proj", (96, 384))):
            # plain heavy-tailed weights: every group spans zero, as the converter's range assertions expect
            w = torch.randn(*shape, generator=g) * 0.02
            w = torch.where(torch.rand(*shape, generator=g) < 0.005, w * 6, w).to(torch.float16)
            q = _build_quantizer(ref, case)
            r = q.quantize(w, return_with_dequant=True, return_with_quant=True)
            arrays[f"{tag}|in|{name}.weight"] = _np(w)
            state[f"{name}.weight"] = r.data
            scales.update(r.scale.state_dict(f"{name}.weight.scale"))
            zero_name = "scaled_zero" if case["zero"] == "PostScale" else "zero"
            scales[f"{name}.weight.{zero_name}"] = r.zero if r.zero.ndim else r.zero.view(1)
        state["layers.0.norm.weight"] = torch.ones(256, dtype=torch.float16)
        out = convert_to_qserve_state_dict(state, scales, bits)
        for k, v in state.items():
            arrays[f"{tag}|model|{k}"] = _np(v)
        for k, v in scales.items():
            arrays[f"{tag}|scale|{k}"] = _np(v)
        for k, v in out.items():
            arrays[f"{tag}|out|{k}"] = v.numpy().copy() if v.dtype in (torch.int8, torch.uint8) else _np(v)
        meta[tag] = dict(case={k: v for k, v in case.items()}, bits=bits,
                         out={k: [str(v.dtype), list(v.shape)] for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "qserve.npz"), **arrays)
    with open(os.path.join(OUT, "qserve.json"), "w") as f:
        json.dump(meta, f, indent=1)


@torch.no_grad()
def gen_nunchaku(ref) -> None:
    """The reference's Nunchaku converter (backend/nunchaku/utils.py:16-352, convert.py:12-72): raw 4- and 8-bit
    packs, scale / micro-scale / low-rank packs, and whole-linear conversions (int4 g64 with low-rank branch, smoothing
    and shift; per-channel int4; fp4 with per-tensor scale and fp8 g16 micro scales; W8A8)."""
    from deepcompressor.backend.nunchaku.convert import convert_to_nunchaku_w4x4y16_linear_state_dict
    from deepcompressor.backend.nunchaku.utils import (NunchakuWeightPacker,
                                                       convert_to_nunchaku_w8x8y16_linear_weight)

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(31337)

    def put(key, t):
        if t.dtype == torch.float8_e4m3fn:
            arrays[key] = t.view(torch.uint8).numpy().copy()
        elif t.dtype in (torch.int8, torch.uint8, torch.int32):
            arrays[key] = t.numpy().copy()
        else:
            arrays[key] = _np(t)

    for bits, shape in ((4, (256, 384)), (8, (128, 192))):
        half = 1 << (bits - 1)
        codes = torch.randint(-half, half, shape, generator=g, dtype=torch.int32)
        arrays[f"pack{bits}|codes"] = codes.numpy().copy()
        arrays[f"pack{bits}|packed"] = NunchakuWeightPacker(bits=bits).pack_weight(codes.clone()).numpy().copy()
    packer = NunchakuWeightPacker(bits=4)
    for dt in ("f16", "bf16"):
        s = (torch.rand(256, 1, 6, 1, generator=g) + 0.01).to(TORCH_DT[dt])
        put(f"scale|{dt}|in", s)
        put(f"scale|{dt}|g64", packer.pack_scale(s, group_size=64))
        put(f"scale|{dt}|chn", packer.pack_scale(s[:, :, :1].contiguous(), group_size=-1))
        ms = (torch.rand(128, 1, 8, 1, generator=g) * 400).to(TORCH_DT[dt])
        put(f"micro|{dt}|in", ms)
        put(f"micro|{dt}|g16", packer.pack_scale(ms, group_size=16))
        down, up = torch.randn(24, 200, generator=g).to(TORCH_DT[dt]), torch.randn(136, 24, generator=g).to(TORCH_DT[dt])
        put(f"lowrank|{dt}|down", down), put(f"lowrank|{dt}|up", up)
        pd = packer.pack_lowrank_weight(packer.pad_lowrank_weight(down, down=True), down=True)
        pu = packer.pack_lowrank_weight(packer.pad_lowrank_weight(up, down=False), down=False)
        put(f"lowrank|{dt}|down_packed", pd), put(f"lowrank|{dt}|up_packed", pu)
        put(f"lowrank|{dt}|down_unpacked", packer.unpack_lowrank_weight(pd, down=True))
        put(f"lowrank|{dt}|up_unpacked", packer.unpack_lowrank_weight(pu, down=False))

    def linear(tag, *, oc, ic, dt, gs, rank=0, with_bias=False, with_smooth=False, with_shift=False, fused=False,
               float_point=False, micro=False):
        dtype = TORCH_DT[dt]
        w = torch.randn(oc, ic, generator=g) * 0.02
        w = torch.where(torch.rand(oc, ic, generator=g) < 0.01, w * 5, w).to(dtype)
        if float_point:
            case = dict(dtype="sfp4_e2m1_all", zero=None, groups=((-1, -1), (1, 16)), sdt=("f16", "sfp8_e4m3_nan"))
            if dt == "bf16":
                case["sdt"] = ("bf16", "sfp8_e4m3_nan")
        else:
            case = dict(dtype="sint4", zero=None, groups=((1, gs),), sdt=(dt,))
        r = _build_quantizer(ref, case).quantize(w, return_with_dequant=True, return_with_quant=True)
        scales = r.scale.state_dict("s")
        kwargs = dict(weight=r.data.clone(), scale=scales["s.0"], float_point=float_point)
        if micro:
            kwargs["subscale"] = scales["s.1"]
        if with_bias:
            kwargs["bias"] = torch.randn(oc, generator=g).to(dtype)
        if with_smooth:
            kwargs["smooth"] = (torch.rand(ic, generator=g) + 0.5).to(dtype)
            kwargs["smooth_fused"] = fused
        if with_shift:
            kwargs["shift"] = torch.randn(ic, generator=g).to(dtype) * 0.1
        if rank:
            kwargs["lora"] = ((torch.randn(rank, ic, generator=g) * 0.05).to(dtype),
                              (torch.randn(oc, rank, generator=g) * 0.05).to(dtype))
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                put(f"{tag}|in|{k}", v)
            elif isinstance(v, tuple):
                put(f"{tag}|in|lora_down", v[0]), put(f"{tag}|in|lora_up", v[1])
        out = convert_to_nunchaku_w4x4y16_linear_state_dict(**kwargs)
        for k, v in out.items():
            put(f"{tag}|out|{k}", v)
        meta[tag] = dict(oc=oc, ic=ic, dtype=dt, float_point=float_point, smooth_fused=fused,
                         out={k: [str(v.dtype), list(v.shape)] for k, v in out.items()})

    linear("w4_g64_branch", oc=200, ic=320, dt="bf16", gs=64, rank=24, with_bias=True, with_smooth=True, with_shift=True)
    linear("w4_g64_fused", oc=128, ic=256, dt="f16", gs=64, rank=16, with_smooth=True, fused=True)
    linear("w4_g64_plain", oc=256, ic=128, dt="f16", gs=64)
    linear("w4_chn", oc=128, ic=128, dt="bf16", gs=128)
    linear("w4_fp4_micro", oc=136, ic=192, dt="bf16", gs=16, rank=8, float_point=True, micro=True)

    for dt in ("f16", "bf16"):
        dtype = TORCH_DT[dt]
        w = (torch.randn(130, 200, generator=g) * 0.02).to(dtype)
        r = _build_quantizer(ref, dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=(dt,))).quantize(
            w, return_with_dequant=True, return_with_quant=True)
        bias = torch.randn(130, generator=g).to(dtype)
        put(f"w8|{dt}|in|weight", r.data), put(f"w8|{dt}|in|scale", r.scale.data), put(f"w8|{dt}|in|bias", bias)
        qw, qs, qb = convert_to_nunchaku_w8x8y16_linear_weight(r.data.clone(), r.scale.data, bias)
        put(f"w8|{dt}|out|weight", qw), put(f"w8|{dt}|out|scale", qs), put(f"w8|{dt}|out|bias", qb)
        meta[f"w8|{dt}"] = dict(out={"weight": [str(qw.dtype), list(qw.shape)], "scale": [str(qs.dtype), list(qs.shape)],
                                     "bias": [str(qb.dtype), list(qb.shape)]})
    np.savez_compressed(os.path.join(OUT, "nunchaku.npz"), **arrays)
    with open(os.path.join(OUT, "nunchaku.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_llm_layer_stages(ref) -> None:
    """The reference's rotation and smoothing stages (app/llm/quant/rotate.py:25-196, app/llm/quant/smooth.py:23-167)
    run on a two-layer random Llama (bf16, grouped-query attention): every parameter before and after the rotation,
    the first layer's cached activations, and the smoothing scales and parameters after the QoQ-style manual smoothing
    and after a grid search on the output / down projections."""
    import types

    from deepcompressor.app.llm.nn.struct import LlmModelStruct
    from deepcompressor.app.llm.quant.quantizer.config import LlmActivationQuantizerConfig, LlmWeightQuantizerConfig
    from deepcompressor.app.llm.quant.rotate import rotate_llm
    from deepcompressor.app.llm.quant.smooth import smooth_llm_layer
    from deepcompressor.calib.config import SearchBasedCalibStrategy
    from deepcompressor.calib.config.rotation import QuantRotationConfig
    from deepcompressor.calib.config.smooth import (
        SkipBasedSmoothCalibConfig,
        SmoothAttentionCalibConfig,
        SmoothSpanMode,
        SmoothTransfomerConfig,
    )
    from deepcompressor.data.cache import IOTensorsCache, TensorCache
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.utils.reshape import LinearReshapeFn
    from deepcompressor.data.zero import ZeroPointDomain
    from transformers import LlamaConfig, LlamaForCausalLM

    arrays, meta = {}, dict(model=TINY_LLAMA)
    bits = lambda t: t.detach().contiguous().view(torch.int16).numpy().copy()  # noqa: E731  (bf16 bit patterns)

    def put_state(tag, model):
        for k, v in model.state_dict().items():
            arrays[f"{tag}|{k}"] = bits(v) if v.dtype == torch.bfloat16 else v.numpy().copy()

    torch.manual_seed(0)
    model = LlamaForCausalLM(LlamaConfig(**TINY_LLAMA, attn_implementation="eager")).to(torch.bfloat16).eval()
    # norms away from one and channels of unequal size, so that folding and smoothing have something to do
    g = torch.Generator().manual_seed(11)
    for name, p in model.named_parameters():
        if "layernorm" in name or name.endswith("norm.weight"):
            p.data = (torch.rand(p.shape, generator=g) + 0.5).to(torch.bfloat16)
    model.model.embed_tokens.weight.data = (model.model.embed_tokens.weight.data.float()
                                            * torch.exp(torch.randn(64, generator=g))).to(torch.bfloat16)
    put_state("init", model)
    tokens = [torch.randint(0, 128, (2, 16), generator=torch.Generator().manual_seed(s)) for s in (1, 2)]
    arrays["tokens"] = torch.stack(tokens).numpy()
    with torch.no_grad():
        arrays["logits|init"] = model(tokens[0]).logits.float().numpy()
    struct = LlmModelStruct.construct(model)
    rotation = rotate_llm(struct, QuantRotationConfig(random=False, transforms=["out_proj"]))
    arrays["rotation"] = rotation.numpy().copy()
    put_state("rotated", model)
    with torch.no_grad():
        arrays["logits|rotated"] = model(tokens[0]).logits.float().numpy()
    layer = struct.backbone_struct.layer_structs[0]
    attn, ffn = layer.attn_struct, layer.ffn_struct
    store: dict[str, dict] = {}

    def grab(name, module):
        def pre(mod, args, kwargs):
            d = store.setdefault(name, dict(inputs=[], outputs=[], kwargs=None))
            d["inputs"].append((args[0] if args else kwargs["hidden_states"]).detach().clone())
            if d["kwargs"] is None:
                d["kwargs"] = {k: v for k, v in kwargs.items() if k != "hidden_states"}

        def post(mod, args, out):
            store[name]["outputs"].append((out[0] if isinstance(out, tuple) else out).detach().clone())

        return [module.register_forward_pre_hook(pre, with_kwargs=True), module.register_forward_hook(post)]

    hooks = []
    watched = [(attn.name, attn.module), (attn.q_proj_name, attn.q_proj), (attn.k_proj_name, attn.k_proj),
               (attn.o_proj_name, attn.o_proj), (ffn.name, ffn.module), (ffn.down_proj_names[0], ffn.down_projs[0]),
               (attn.v_proj_name, attn.v_proj), (ffn.up_proj_names[0], ffn.up_projs[0]),
               (ffn.up_proj_names[1], ffn.up_projs[1])]
    for name, mod in watched:
        hooks += grab(name, mod)
    with torch.no_grad():
        for t in tokens:
            model(t)
    for h in hooks:
        h.remove()
    for name, d in store.items():
        arrays[f"acts|{name}|inputs"] = np.stack([bits(x) for x in d["inputs"]])
        arrays[f"acts|{name}|outputs"] = np.stack([bits(x) for x in d["outputs"]])
    meta["watched"] = [name for name, _ in watched]
    # what the attention block takes besides the hidden states (the same for every batch: same length, no padding)
    captured = store[attn.name]["kwargs"]
    layer_kwargs = dict(attention_mask=captured["attention_mask"], position_embeddings=captured["position_embeddings"])
    mask, (cos, sin) = layer_kwargs["attention_mask"], layer_kwargs["position_embeddings"]
    arrays["kwargs|attention_mask"] = mask.float().numpy()
    arrays["kwargs|cos"], arrays["kwargs|sin"] = cos.float().numpy(), sin.float().numpy()
    meta["kwargs_dtypes"] = dict(attention_mask=str(mask.dtype), cos=str(cos.dtype), sin=str(sin.dtype))

    def tc(ts):
        return TensorCache([t.clone() for t in ts], channels_dim=-1, reshape=LinearReshapeFn(), num_cached=len(ts),
                           num_total=len(ts), num_samples=len(ts) * ts[0].shape[0])

    layer_cache = {name: IOTensorsCache(inputs=tc(d["inputs"]), outputs=tc(d["outputs"])) for name, d in store.items()}
    wg = LlmWeightQuantizerConfig(dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale,
                                  group_shapes=((1, 16),), scale_dtypes=(torch.float16,))
    ip = LlmActivationQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None, group_shapes=((1, -1),),
                                      scale_dtypes=(torch.float16,))
    op = LlmActivationQuantizerConfig(dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale,
                                      group_shapes=((1, 16),), scale_dtypes=(torch.float16,), skips=["attn_q"])
    spans = [(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)]
    rotated = {k: v.clone() for k, v in model.state_dict().items()}
    plans = {
        "manual": (SkipBasedSmoothCalibConfig(strategy=SearchBasedCalibStrategy.Manual, spans=spans, alpha=0.3, beta=0.7,
                                              skips=[]),
                   SmoothAttentionCalibConfig(strategy=SearchBasedCalibStrategy.Manual, spans=spans, alpha=0.5, beta=0)),
        "grid": (SkipBasedSmoothCalibConfig(strategy=SearchBasedCalibStrategy.GridSearch, spans=spans, alpha=0.5, beta=-2,
                                            num_grids=4, skips=["qkv_proj", "up_proj"]), None),
    }
    for tag, (proj, att) in plans.items():
        model.load_state_dict(rotated)
        cfg = types.SimpleNamespace(wgts=wg, ipts=ip, opts=op, enabled_wgts=True, enabled_ipts=True, enabled_opts=True,
                                    smooth=SmoothTransfomerConfig(proj=proj, attn=att), develop_dtype=torch.float32)
        cache: dict[str, torch.Tensor] = {}
        smooth_llm_layer(layer, cfg, cache, layer_cache=layer_cache, layer_kwargs=layer_kwargs)
        for k, v in cache.items():
            arrays[f"smooth|{tag}|scale|{k}"] = v.float().numpy().copy()
        put_state(f"smooth|{tag}|state", model)
        with torch.no_grad():
            arrays[f"logits|smooth|{tag}"] = model(tokens[0]).logits.float().numpy()
        meta[f"smooth|{tag}"] = dict(scales=sorted(cache))
    # channel reorder inside the layer (app/llm/quant/reorder.py:48-236): the output projection's order with grouped
    # query heads (also permuting v_proj's rows) and the down projection's (permuting up / gate rows)
    from deepcompressor.app.llm.quant.reorder import reorder_llm_layer
    from deepcompressor.calib.config.reorder import ChannelOrderCalibConfig, SkipBasedChannelOrderConfig

    reorder_plans = {
        "manual": dict(strategy=SearchBasedCalibStrategy.Manual, channel_metric=ChannelOrderCalibConfig.ChannelMetric.InputsAbsMax,
                       channel_index=ChannelOrderCalibConfig.ChannelIndex.Sequential),
    }
    for tag, kw in reorder_plans.items():
        model.load_state_dict(rotated)
        cfg = types.SimpleNamespace(wgts=wg, ipts=ip, opts=op, enabled_wgts=True, enabled_ipts=True, enabled_opts=True,
                                    reorder=SkipBasedChannelOrderConfig(skips=["residual"], **kw),
                                    develop_dtype=torch.float32)
        cache = {}
        reorder_llm_layer(layer, cfg, cache, layer_cache=layer_cache, layer_kwargs=layer_kwargs)
        for k, v in cache.items():
            arrays[f"reorder|{tag}|index|{k}"] = v.numpy().copy()
        put_state(f"reorder|{tag}|state", model)
        meta[f"reorder|{tag}"] = dict(indexes=sorted(cache))
    # weight and activation stages on the smoothed layer (app/llm/quant/weight.py:24-105, activation.py:25-192):
    # QoQ-style progressive weights (sint8 per channel, then uint4 groups of 16 with integer scales) with a range
    # search per step; dynamic sint8 inputs per token, uint4 keys / values in groups of 16
    from deepcompressor.app.llm.quant.activation import quantize_llm_layer_activations
    from deepcompressor.app.llm.quant.weight import quantize_llm_layer_weights
    from deepcompressor.calib.config import (
        SearchBasedCalibGranularity,
        SearchBasedCalibObjective,
        SkipBasedDynamicRangeCalibConfig,
    )

    model.load_state_dict(rotated)
    cfg = types.SimpleNamespace(wgts=wg, ipts=ip, opts=op, enabled_wgts=True, enabled_ipts=True, enabled_opts=True,
                                smooth=SmoothTransfomerConfig(proj=plans["manual"][0], attn=plans["manual"][1]),
                                develop_dtype=torch.float32)
    smooth_llm_layer(layer, cfg, {}, layer_cache=layer_cache, layer_kwargs=layer_kwargs)
    calib = SkipBasedDynamicRangeCalibConfig(
        degree=2, objective=SearchBasedCalibObjective.TensorError, strategy=SearchBasedCalibStrategy.GridSearch,
        granularity=SearchBasedCalibGranularity.Group, max_shrink=0.4, max_expand=1.0, num_grids=8, skips=[])
    qoq = LlmWeightQuantizerConfig(dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale,
                                   group_shapes=((1, -1), (1, 16)), scale_dtypes=(torch.float16, QuantDataType.from_str("sint8")),
                                   intermediate_dtypes=(QuantDataType.from_str("sint8"),), intermediate_levels=(0,),
                                   calib_range=calib)
    cfg.wgts = qoq
    weight_ranges: dict = {}
    scale_sd = quantize_llm_layer_weights(layer, cfg, weight_ranges, layer_cache=layer_cache, layer_kwargs=layer_kwargs,
                                          return_with_scale_state_dict=True)
    for k, v in scale_sd.items():
        arrays[f"chain|scale|{k}"] = _np(v if v.ndim else v.view(1))
    put_state("chain|state", model)
    act_ranges: dict = {}
    quantize_llm_layer_activations(layer, cfg, act_ranges, layer_cache=layer_cache, layer_kwargs=layer_kwargs)
    with torch.no_grad():
        arrays["logits|chain"] = model(tokens[0]).logits.float().numpy()
    meta["chain"] = dict(scales=sorted(scale_sd), weight_ranges=sorted(weight_ranges), act_ranges=sorted(act_ranges),
                         weight_ranges_in_order=list(weight_ranges))
    # the weight stage with the GPTQ kernel on the rotated layer (gptq.yaml's settings, groups of 16): scored in the
    # tests by codes / layer outputs, not bit for bit (GPTQ is not reproducible across BLAS / LAPACK builds)
    from deepcompressor.quantizer.kernel.gptq import QuantGptqConfig

    model.load_state_dict(rotated)
    cfg.wgts = LlmWeightQuantizerConfig(
        dtype=QuantDataType.from_str("uint4"), zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 16),),
        scale_dtypes=(torch.float16,), calib_range=calib,
        kernel_gptq=QuantGptqConfig(damp_percentage=0.01, block_size=128, num_inv_tries=250, hessian_block_size=512))
    gptq_ranges: dict = {}
    gptq_scales = quantize_llm_layer_weights(layer, cfg, gptq_ranges, layer_cache=layer_cache, layer_kwargs=layer_kwargs,
                                             return_with_scale_state_dict=True)
    for k, v in gptq_scales.items():
        arrays[f"gptq|scale|{k}"] = _np(v if v.ndim else v.view(1))
    put_state("gptq|state", model)
    meta["gptq"] = dict(scales=sorted(gptq_scales))
    # static per-tensor sint8 inputs with a searched range (smoothquant-static.yaml's kind of setting): every range is
    # scored on the outputs of the module the reference evaluates it on (attention block, the linear, the mlp)
    model.load_state_dict(rotated)
    static_calib = SkipBasedDynamicRangeCalibConfig(
        degree=2, objective=SearchBasedCalibObjective.OutputsError, strategy=SearchBasedCalibStrategy.GridSearch,
        granularity=SearchBasedCalibGranularity.Layer, max_shrink=0.4, max_expand=1.0, num_grids=4, skips=[])
    cfg.ipts = LlmActivationQuantizerConfig(dtype=QuantDataType.from_str("sint8"), zero_point=None,
                                            group_shapes=((-1, -1),), scale_dtypes=(torch.float16,), static=True,
                                            calib_range=static_calib)
    cfg.enabled_opts, cfg.opts = False, LlmActivationQuantizerConfig(dtype=None)
    static_ranges: dict = {}
    quantize_llm_layer_activations(layer, cfg, static_ranges, layer_cache=layer_cache, layer_kwargs=layer_kwargs)
    for k, v in static_ranges.items():
        (rng,) = v["dynamic_range"]
        arrays[f"static|max|{k}"] = _np(rng["max"].reshape(-1))
    with torch.no_grad():
        arrays["logits|static"] = model(tokens[0]).logits.float().numpy()
    meta["static"] = dict(ranges=sorted(static_ranges))
    np.savez_compressed(os.path.join(OUT, "llm_layer_stages.npz"), **arrays)
    with open(os.path.join(OUT, "llm_layer_stages.json"), "w") as f:
        json.dump(meta, f, indent=1)


@torch.no_grad()
def gen_patch_linear(ref) -> None:
    """The reference's linear patches (nn/patch/linear.py:10-128): a linear split over a concatenated input, and a
    linear with a shifted input (scalar shift; the per-channel form of the reference's from_linear cannot run)."""
    import torch.nn as nn
    from deepcompressor.nn.patch.linear import ConcatLinear, ShiftedLinear

    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(606)
    for dt in ("f32", "bf16"):
        lin = nn.Linear(96, 40, bias=True).to(TORCH_DT[dt])
        lin.weight.data = (torch.randn(40, 96, generator=g) * 0.1).to(TORCH_DT[dt])
        lin.bias.data = (torch.randn(40, generator=g) * 0.1).to(TORCH_DT[dt])
        x = torch.randn(3, 5, 96, generator=g).to(TORCH_DT[dt])
        arrays[f"{dt}|w"], arrays[f"{dt}|b"], arrays[f"{dt}|x"] = _np(lin.weight.data), _np(lin.bias.data), _np(x)
        cat = ConcatLinear.from_linear(lin, [32, 16])
        arrays[f"{dt}|concat|y"] = _np(cat(x))
        for i, part in enumerate(cat.linears):
            arrays[f"{dt}|concat|w{i}"] = _np(part.weight.data)
        meta[f"{dt}|concat"] = dict(widths=cat.in_features_list, bias_on=[p.bias is not None for p in cat.linears])
        sh = ShiftedLinear.from_linear(lin, 0.17)
        arrays[f"{dt}|shift|bias"], arrays[f"{dt}|shift|y"] = _np(sh.linear.bias.data), _np(sh(x))
        arrays[f"{dt}|shift|shift"] = _np(sh.shift)
    # the convolution counterparts (nn/patch/conv.py:14-182): zero padding, no padding with a per-block shift
    from deepcompressor.nn.patch.conv import ConcatConv2d, ShiftedConv2d
    for dt in ("f32", "bf16"):
        conv = nn.Conv2d(12, 10, 3, stride=1, padding=1, bias=True).to(TORCH_DT[dt])
        conv.weight.data = (torch.randn(10, 12, 3, 3, generator=g) * 0.1).to(TORCH_DT[dt])
        conv.bias.data = (torch.randn(10, generator=g) * 0.1).to(TORCH_DT[dt])
        x = torch.randn(2, 12, 7, 6, generator=g).to(TORCH_DT[dt])
        arrays[f"{dt}|conv|w"], arrays[f"{dt}|conv|b"], arrays[f"{dt}|conv|x"] = _np(conv.weight.data), _np(conv.bias.data), _np(x)
        cat = ConcatConv2d.from_conv2d(conv, [4, 5])
        arrays[f"{dt}|conv|concat|y"] = _np(cat(x))
        meta[f"{dt}|conv|concat"] = dict(widths=cat.in_channels_list, bias_on=[p.bias is not None for p in cat.convs])
        sh = ShiftedConv2d.from_conv2d(conv, 0.21)
        arrays[f"{dt}|conv|shift|bias"], arrays[f"{dt}|conv|shift|y"] = _np(sh.conv.bias.data), _np(sh(x))
        bare = nn.Conv2d(12, 10, 3, stride=2, padding=0, bias=False).to(TORCH_DT[dt])
        bare.weight.data = conv.weight.data.clone()
        block_shift = torch.tensor([0.1, -0.2, 0.3])
        # (from_conv2d cannot take this: it passes the padding as a tuple, which the constructor compares with 0)
        sh = ShiftedConv2d(12, 10, 3, shift=block_shift, stride=2, padding=0, bias=True, dtype=TORCH_DT[dt])
        sh.conv.weight.data.copy_(bare.weight.data)
        sh.conv.bias.data.zero_()
        arrays[f"{dt}|conv|blockshift|y"] = _np(sh(x))
        arrays[f"{dt}|conv|blockshift|shift"] = _np(sh.shift)
    np.savez_compressed(os.path.join(OUT, "patch_linear.npz"), **arrays)
    with open(os.path.join(OUT, "patch_linear.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_ste(ref) -> None:
    """Gradients of the reference's simple_quantize for tensors that require grad (quantizer/impl/simple.py:22-51 with
    quantizer/impl/ste.py): rounding passes the gradient through, the clamps mask it."""
    from deepcompressor.data.dtype import QuantDataType
    from deepcompressor.data.range import QuantRange
    from deepcompressor.quantizer.impl.simple import simple_quantize

    cases = {
        "sint4": dict(dtype="sint4", zp=False, amp=4.0),
        "sint4_zp": dict(dtype="sint4", zp=True, amp=4.0),
        "uint4_zp": dict(dtype="uint4", zp=True, amp=6.0, shift=7.0),
        "sint8_range": dict(dtype="sint8", zp=False, amp=80.0, qr=(-119.0, 119.0)),
        "sint4_delta": dict(dtype="sint4", zp=True, amp=4.0, delta=True),  # floor + a given 0/1 offset
        "sfp4": dict(dtype="sfp4_e2m1_all", zp=False, amp=3.5),
        # (a float type with a quant_range cannot run with grad in the reference: clamp_ on a view made inside
        # the custom Function is refused by autograd)
        "sfp8": dict(dtype="sfp8_e4m3_nan", zp=False, amp=300.0),
        "uexp4": dict(dtype="uexp4_all", zp=False, amp=1.0, positive=True),
        "f16": dict(dtype="f16", zp=False, amp=2.0),
        "f16_range": dict(dtype="f16", zp=False, amp=2.0, qr=(-1.5, 2.5)),
    }
    arrays, meta = {}, {}
    g = torch.Generator().manual_seed(707)
    for name, c in cases.items():
        x = torch.randn(8, 96, generator=g) * c["amp"] + c.get("shift", 0.0)
        if c.get("positive"):
            x = torch.exp(torch.randn(8, 96, generator=g) * 4.0)
        x[0, :8] = torch.arange(8) - 0.5 if not c.get("positive") else x[0, :8]  # ties and exact bounds
        up = torch.randn(8, 96, generator=g)
        leaf = x.clone().requires_grad_(True)
        qdt = TORCH_DT[c["dtype"]] if c["dtype"] in TORCH_DT else QuantDataType.from_str(c["dtype"])
        qr = QuantRange(*c["qr"]) if "qr" in c else None
        delta = None
        if c.get("delta"):
            delta = (torch.rand(8, 96, generator=g) < 0.5).float()
            arrays[f"{name}|delta"] = _np(delta)
        out = simple_quantize(leaf * 1.0, quant_dtype=qdt, has_zero_point=c["zp"], quant_range=qr, round_delta=delta)
        out.backward(up)
        arrays[f"{name}|x"], arrays[f"{name}|up"] = _np(x), _np(up)
        arrays[f"{name}|out"], arrays[f"{name}|grad"] = _np(out.detach()), _np(leaf.grad)
        meta[name] = {k: v for k, v in c.items()}
    np.savez_compressed(os.path.join(OUT, "ste.npz"), **arrays)
    with open(os.path.join(OUT, "ste.json"), "w") as f:
        json.dump(meta, f, indent=1)


FP_ZERO_CASES = {
    # floating-point types with a zero point: the range is centred on the group mean (data/range.py:323-341)
    "sfp4_post_g32": dict(dtype="sfp4_e2m1_all", zero="PostScale", groups=((1, 32),), sdt=("f16",), shape=(32, 256)),
    "sfp4_pre_g32": dict(dtype="sfp4_e2m1_all", zero="PreScale", groups=((1, 32),), sdt=("f16",), shape=(32, 256)),
    "sfp8_post_gchn": dict(dtype="sfp8_e4m3_nan", zero="PostScale", groups=((1, -1),), sdt=("f16",), shape=(16, 256)),
}


def gen_fp_zero(ref) -> None:
    from deepcompressor.data.range import DynamicRange

    arrays, meta = {}, {}
    for cname, case in FP_ZERO_CASES.items():
        for dt in ("bf16", "f16"):
            for ratio in (None, 0.8):
                g = torch.Generator().manual_seed(900 + len(cname) + len(dt))
                x = (torch.randn(*case["shape"], generator=g) * 0.05 + 0.03 * torch.randn(case["shape"][0], 1, generator=g)
                     ).to(TORCH_DT[dt])  # rows with a non-zero mean, so that the centring matters
                dyn = None if ratio is None else DynamicRange(ratio=ratio)
                q = _build_quantizer(ref, case)
                r = q.quantize(x, return_with_dequant=False, return_with_quant=True, dynamic_range=dyn)
                deq = _build_quantizer(ref, case).quantize(x, dynamic_range=dyn).data
                tag = f"{cname}|{dt}|r{ratio}"
                arrays[f"{tag}|x"], arrays[f"{tag}|codes"], arrays[f"{tag}|dequant"] = _np(x), _np(r.qdata), _np(deq)
                arrays[f"{tag}|scale"], arrays[f"{tag}|zero"] = _np(r.scale.data), _np(r.zero)
                meta[tag] = dict(case=cname, in_dtype=dt, ratio=ratio, codebook_via_restatement=True)
    np.savez_compressed(os.path.join(OUT, "fp_zero.npz"), **arrays)
    with open(os.path.join(OUT, "fp_zero.json"), "w") as f:
        json.dump(dict(cases={k: {kk: (list(vv) if kk == "shape" else vv) for kk, vv in v.items()}
                              for k, v in FP_ZERO_CASES.items()}, entries=meta), f, indent=1)


# Three progressive steps (no example config of the reference uses them; impl/base.py:211-279 handles any number):
# per-row fp16 scales onto sint8, groups of 256 with integer scales onto sint6, groups of 64 onto uint4.
THREE_STEP_CASES = {
    "three_post": dict(dtype="uint4", zero="PostScale", groups=((1, -1), (1, 256), (1, 64)), sdt=("f16", "sint8", "sint8"),
                       inter=("sint8", "sint6"), levels=(0, 1), shape=(32, 512)),
    "three_pre": dict(dtype="uint4", zero="PreScale", groups=((1, -1), (1, 256), (1, 64)), sdt=("f16", "sint8", "sint8"),
                      inter=("sint8", "sint6"), levels=(0, 1), shape=(16, 512)),
    "three_sym": dict(dtype="sint4", zero=None, groups=((1, -1), (1, 128), (1, 32)), sdt=("f16", "sint8", "sint8"),
                      inter=("sint8", "sint6"), levels=(0, 1), shape=(16, 256)),
}


def gen_three_steps(ref) -> None:
    from deepcompressor.data.range import DynamicRange

    meta, arrays = {}, {}
    for cname, case in THREE_STEP_CASES.items():
        for dt in ("bf16", "f16", "f32"):
            for dyn in (None, (0.9, 1.0, 0.8)):
                if dyn is not None and dt != "bf16":
                    continue
                w = _mk_input(case["shape"], dt, seed=len(cname) * 5 + len(dt))
                q = _build_quantizer(ref, case)
                kwargs = {}
                if dyn is not None:
                    kwargs["dynamic_range"] = tuple(DynamicRange(ratio=r) for r in dyn)
                codes = q.quantize(w, return_with_dequant=False, return_with_quant=True, **kwargs)
                deq = q.quantize(w, return_with_dequant=True, return_with_quant=False, **kwargs)
                tag = f"{cname}|{dt}" + ("|ratios" if dyn else "")
                arrays[f"{tag}|x"] = _np(w)
                arrays[f"{tag}|codes"] = _np(codes.qdata)
                arrays[f"{tag}|dequant"] = _np(deq.data)
                arrays[f"{tag}|scale"] = _np(codes.scale.data)
                sd = codes.scale.state_dict("s")
                for lvl, leaf in enumerate(sd.values()):
                    arrays[f"{tag}|leaf{lvl}"] = _np(leaf)
                arrays[f"{tag}|zero"] = _np(codes.zero)
                meta[tag] = dict(case=cname, in_dtype=dt, dyn=list(dyn) if dyn else None,
                                 view_shape=list(codes.view_shape), n_leaves=len(sd))
    np.savez_compressed(os.path.join(OUT, "three_steps.npz"), **arrays)
    with open(os.path.join(OUT, "three_steps.json"), "w") as f:
        json.dump(dict(cases={k: {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in v.items()}
                              for k, v in THREE_STEP_CASES.items()}, entries=meta), f, indent=1)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    which = sys.argv[1:] or ["fake_quant", "range_search", "range_wide", "range_products", "tables", "gptq", "rotate", "smooth", "smooth_attn", "smooth_lowrank", "lowrank", "range_acts", "reorder", "llm_stage", "round_delta", "exp_scale", "qserve", "fp_zero", "nunchaku", "smooth_products", "range_acts_products", "llm_layer_stages", "patch_linear", "ste", "smooth_tensor", "branch", "three_steps"]
    for name in which:
        globals()[f"gen_{name}"](ref)
        print("generated", name)


if __name__ == "__main__":
    main()
