"""CPU-only checks of the host-side mirror of the reference interface (no GPU, no kernels run)."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch
from _product import build_config
from conftest import ROOT, load_golden

from deepcompressor_amd.calib import DynamicRangeCalibConfig, SearchBasedCalibStrategy
from deepcompressor_amd.data.dtype import QDType, QuantDataType
from deepcompressor_amd.data.range import ProtectiveQuantRange, QuantRange
from deepcompressor_amd.data.utils import ShapeUtils
from deepcompressor_amd.data.zero import ZeroPointDomain
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig
from deepcompressor_amd.quantizer.impl.info import QuantInfo

TB = load_golden("tables")
FQ = load_golden("fake_quant")
RS = load_golden("range_search")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dcq.h")).read()
    declared = set(re.findall(r"\b(dcq_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 12
    lib = ctypes.CDLL(os.path.join(ROOT, "deepcompressor_amd", "libdcq.so"))
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, f"libdcq.so lacks {missing}"
    lib.dcq_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dcq_version()


def test_dtypes_and_codebooks_match_reference_tables():
    for key in TB.arrays.files:
        if key.startswith("codebook|"):
            name = key.split("|")[1]
            got = QuantDataType.from_str(name).get_codebook().values.numpy()
            assert np.array_equal(got.view(np.uint32), TB[key].view(np.uint32)), name
    for name, p in TB.meta["dtype_props"].items():
        d = QuantDataType.from_str(name)
        assert (d.min_value, d.max_value) == (p["min"], p["max"]), name
        assert (d.is_exponent, d.is_float_point, d.signed, d.total_bits) == (p["is_exp"], p["is_fp"], p["signed"],
                                                                             p["bits"]), name
        assert (d.max_exponent_value, d.min_exponent_value) == (p["max_exp"], p["min_exp"]), name
        assert QuantDataType.from_str(name) is d
        if not d.is_exponent:  # exponent-only names carry total bits on input, exponent bits on output
            assert str(d) == name
    assert QDType.sint4 is QuantDataType.from_str("sint4")


def test_protective_ranges_match_reference():
    for key, (lo, hi) in TB.meta["protective"].items():
        outer, inner, zd = key.split("|")
        r = ProtectiveQuantRange.construct(QuantDataType.from_str(outer), QuantDataType.from_str(inner),
                                           None if zd == "None" else ZeroPointDomain[zd])
        assert (r.min, r.max) == (lo, hi), key


def test_quant_range_symmetric_without_zero_point():
    assert QuantRange.construct(QDType.sint4, has_zero_point=False).to_dict() == {"min": -7, "max": 7}
    assert QuantRange.construct(QDType.sint4, has_zero_point=True).to_dict() == {"min": -8, "max": 7}
    assert QuantRange.construct(QDType.uint4, has_zero_point=True).to_dict() == {"min": 0, "max": 15}


def test_view_shapes_match_fixtures():
    for tag, e in FQ.meta["entries"].items():
        case = FQ.meta["cases"][e["case"]]
        cfg = build_config(case)
        info = QuantInfo.construct(cfg.decompose(), torch.Size(case["shape"]), torch.float16)
        assert list(info.steps[-1].tensor_view_shape) == e["view_shape"], tag


def test_progressive_decomposition_qoq():
    cfg = build_config(FQ.meta["cases"]["qoq_g128"])
    dec = cfg.decompose()
    assert dec.num_steps == 2
    assert dec.steps[0].dtype == QDType.sint8 and dec.steps[0].zero_point is None
    assert dec.steps[1].dtype == QDType.uint4 and dec.steps[1].scale_dtypes == (QDType.sint8,)
    info = QuantInfo.construct(dec, torch.Size((64, 512)), torch.float16)
    assert info.steps[0].tensor_quant_range.to_dict() == {"min": -120, "max": 120}
    assert info.steps[1].range_bound.to_dict() == {"min": -120, "max": 120}


def test_group_shape_names_and_dirnames():
    assert ShapeUtils.infer_group_shape_name((1, 128, -1)) == "g128"
    assert ShapeUtils.infer_group_shape_name((1, -1, -1)) == "gchn"
    assert ShapeUtils.infer_group_shape_name((-1, -1, -1)) == "tsnr"
    assert ShapeUtils.infer_group_shape_name((-1, 64, -1)) == "tsg64"
    assert ShapeUtils.infer_group_shape_name((4, -1, -1)) == "t4gchn"
    assert ShapeUtils.infer_group_shape_name((1, 16, 1, 1)) == "v16"
    assert ShapeUtils.infer_group_shape_name((1, 16, 2, 2)) == "1x16x2x2"
    cfg = QuantizerConfig(dtype=QDType.uint4, zero_point=ZeroPointDomain.PostScale, group_shapes=((1, 128),),
                          scale_dtypes=(torch.float16,))
    assert cfg.generate_dirnames() == ["4", "uint4.zp", "g128.fp16"]
    qoq = build_config(FQ.meta["cases"]["qoq_g128"])
    assert qoq.generate_dirnames() == ["4", "uint4.zp", "g128.sint8.gchn.fp16"]


def test_search_ratio_grid_is_the_reference_grid():
    for tag, m in RS.meta.items():
        cfg = DynamicRangeCalibConfig(strategy=SearchBasedCalibStrategy.GridSearch, num_grids=m["num_grids"])
        flat = [r for it in cfg.get_ratios() for r in it]
        assert np.array_equal(np.asarray(flat, dtype=np.float64), RS[f"{tag}|ratios"]), tag


def test_state_dict_layout_roundtrip():
    q = Quantizer(QuantizerConfig(dtype=QDType.sint4, group_shapes=((1, 128),), scale_dtypes=(torch.float16,)),
                  channels_dim=None)
    from deepcompressor_amd.data.range import DynamicRange
    q.dynamic_range = DynamicRange(max=torch.ones(4, 1, 2, 1))
    sd = q.state_dict()
    assert set(sd) == {"channels_dim", "scale", "zero", "dynamic_range", "range_bound", "quant_range"}
    assert set(sd["dynamic_range"]) == {"min", "max", "ratio"}
    q2 = Quantizer(q.config)
    q2.load_state_dict(sd)
    assert torch.equal(q2.dynamic_range.max, q.dynamic_range.max) and q2.dynamic_range.min is None


def test_product_has_no_cpu_path():
    q = Quantizer(QuantizerConfig(dtype=QDType.sint4, group_shapes=((1, 128),), scale_dtypes=(torch.float16,)))
    with pytest.raises(Exception, match="GPU"):
        q.quantize(torch.zeros(2, 128, dtype=torch.float16))


def test_llm_stage_config_names_match_reference():
    """Cache directory names of the weight-stage config (app/llm/quant/quantizer/config.py:78-98 and the quantizer
    config's generate_dirnames) as the reference produced them."""
    from _stage import stage_config
    from conftest import load_golden

    LS = load_golden("llm_stage")
    for tag, m in LS.meta.items():
        cfg = stage_config(m["case"])
        assert cfg.static is True
        assert cfg.generate_calib_dirname() == m["calib_dirname"], tag
        assert cfg.generate_dirnames(prefix="w") == m["dirnames"], tag


def test_rotation_refuses_tied_embeddings():
    transformers = pytest.importorskip("transformers")
    from deepcompressor_amd.app import rotate_llm
    from deepcompressor_amd.calib.config import QuantRotationConfig

    model = transformers.LlamaForCausalLM(transformers.LlamaConfig(
        vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
        num_key_value_heads=1, tie_word_embeddings=True))
    dtype = model.lm_head.weight.dtype
    with pytest.raises(ValueError, match="tied"):
        rotate_llm(model, QuantRotationConfig())
    assert model.lm_head.weight.dtype == dtype  # nothing was touched


def test_stage_checkpoint_files_round_trip(tmp_path):
    """rotation.pt / reorder.pt / smooth.pt / branch.pt / wgts.pt / acts.pt / scale.pt / model.pt in the reference's
    shapes (app/llm/ptq.py:74-78, 121-306; app/diffusion/ptq.py:205-222): written one torch.save per file, read back with tensor-only loading."""
    import torch

    from deepcompressor_amd.app import load_checkpoint, load_weight_checkpoint, save_checkpoint

    g = torch.Generator().manual_seed(0)
    parts = dict(
        rotation=torch.randn(8, 8, generator=g, dtype=torch.float64),
        reorder={"model.layers.0.self_attn.o_proj": torch.randperm(8, generator=g)},
        smooth={"model.layers.0.mlp.down_proj": torch.rand(8, generator=g) + 0.5},
        wgts={"model.layers.0.mlp.up_proj": {"range_bound": None, "dynamic_range": [{"min": None, "max": torch.rand(4, 1, 2, 1, generator=g), "ratio": None}]}},
        acts={"model.layers.0.mlp.up_proj.input": {"dynamic_range": [{"min": None, "max": torch.rand(1), "ratio": 0.9}]}},
        scale={"model.layers.0.mlp.up_proj.weight.scale.0": torch.rand(4, 1, 2, 1, generator=g),
               "model.layers.0.mlp.up_proj.weight.zero": 0},
        model={"model.layers.0.mlp.up_proj.weight": torch.randn(4, 8, generator=g).to(torch.bfloat16)},
        branch={"transformer_blocks.0.attn.to_q": {"a.weight": torch.randn(2, 8, generator=g).to(torch.bfloat16),
                                                   "b.weight": torch.randn(12, 2, generator=g).to(torch.bfloat16)}},
    )
    written = save_checkpoint(str(tmp_path), **parts)
    assert sorted(written) == sorted(f"{k}.pt" for k in parts)
    back = load_checkpoint(str(tmp_path))

    def same(a, b):
        if isinstance(a, torch.Tensor):
            return isinstance(b, torch.Tensor) and a.dtype == b.dtype and torch.equal(a, b)
        if isinstance(a, dict):
            return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, list):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b

    for name, obj in parts.items():
        assert same(obj, back[f"{name}.pt"]), name
    assert back["rotation.pt"].dtype == torch.float64
    assert sorted(load_weight_checkpoint(str(tmp_path))) == ["model.pt", "scale.pt", "wgts.pt"]
    assert save_checkpoint(str(tmp_path / "empty")) == {} and load_checkpoint(str(tmp_path / "empty")) == {}
