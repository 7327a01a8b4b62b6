"""Smooth-quantization oracle and host-side configuration against fixtures produced by the reference
(oracle/make_golden.py smooth).  CPU only."""

import numpy as np
import pytest
from conftest import assert_bit_equal, load_golden

import quant_oracle as qo
import smooth_oracle as so
from deepcompressor_amd.calib.config import (SearchBasedCalibStrategy, SmoothCalibConfig, SmoothSpanMode)

G = load_golden("smooth")


def _num(text):
    return float(text) if "." in text else int(text)


@pytest.mark.parametrize("key", sorted(G.meta["pairs"]))
def test_alpha_beta_candidates_and_dirnames(key):
    parts = key.split("|")
    if parts[0] == "manual":
        cfg = SmoothCalibConfig(spans=[("AbsMax", "RootMeanSquare")], alpha=_num(parts[1]), beta=_num(parts[2]))
        names = cfg.generate_dirnames()
    else:
        cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=_num(parts[0]),
                                beta=_num(parts[1]), num_grids=5, strategy=SearchBasedCalibStrategy.GridSearch)
        names = cfg.generate_dirnames(prefix="smooth")
    assert [list(p) for p in cfg.get_alpha_beta_pairs()] == G.meta["pairs"][key]
    assert names == G.meta["dirnames"][key]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_oracle_spans_and_scales(dt):
    xs = [G[f"span|{dt}|x{i}"] for i in range(3)]
    ws = [G[f"span|{dt}|w{i}"] for i in range(2)]
    conv = G[f"span|{dt}|conv"]
    for name, tensors in (("x", xs), ("w", ws), ("conv", conv)):
        assert_bit_equal(so.channel_metric(tensors, 96, "abs_max"), G[f"span|{dt}|{name}|abs_max"], f"{name} abs_max")
        for mode in ("abs_mean", "root_mean_square"):  # float32 summation order differs: 4 ulp
            want = G[f"span|{dt}|{name}|{mode}"]
            got = so.channel_metric(tensors, 96, mode)
            assert np.all(np.abs(got - want) <= 4 * np.spacing(np.abs(want))), (name, mode)
    a, b = G[f"span|{dt}|x|abs_max"], G[f"span|{dt}|w|abs_max"]
    for alpha, beta in ((0.5, 0.5), (0.3, 0.7), (0.85, 0), (0, 0.4), (1.0, 1.0)):
        want = G[f"scale|{dt}|{alpha}|{beta}"]
        got = so.smooth_scale(a, b, alpha, beta)
        assert np.all(np.abs(got - want) <= 2 * np.spacing(np.abs(want))), (alpha, beta)  # float32 pow: <= 1 ulp each
        assert (got == 1).sum() == (want == 1).sum()


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_oracle_apply(dt):
    s = G[f"apply|{dt}|scale"]
    for i in range(2):
        assert_bit_equal(so.upscale(G[f"apply|{dt}|w{i}"], dt, s, 1), G[f"apply|{dt}|w{i}_out"], f"w{i}")
    assert_bit_equal(so.downscale(G[f"apply|{dt}|prev_w"], dt, s, 0), G[f"apply|{dt}|prev_w_out"], "prev w")
    assert_bit_equal(so.downscale(G[f"apply|{dt}|prev_b"], dt, s, 0), G[f"apply|{dt}|prev_b_out"], "prev b")
    s2 = G[f"apply_gqa|{dt}|scale"]
    assert_bit_equal(so.upscale(G[f"apply_gqa|{dt}|w"], dt, s2, 1), G[f"apply_gqa|{dt}|w_out"], "gqa w")
    assert_bit_equal(so.downscale(G[f"apply_gqa|{dt}|prev_w"], dt, so.downscale_for_producer(s2, 4, 2), 0),
                     G[f"apply_gqa|{dt}|prev_w_out"], "gqa prev")


def test_oracle_search_errors():
    """Every candidate's outputs error of the reference's grid search, and its choice."""
    m = G.meta["search|grid"]
    cfg = SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=m["alpha"], beta=m["beta"],
                            num_grids=20, strategy=SearchBasedCalibStrategy.GridSearch)
    pairs = cfg.get_alpha_beta_pairs()
    assert len(pairs) == m["num_candidates"]
    weights = [G["search|grid|wa"], G["search|grid|wb"]]
    xs = [G[f"search|grid|x{i}"] for i in range(3)]
    a_span = so.channel_metric([x.reshape(-1, 64) for x in xs], 64, "abs_max")
    b_span = so.channel_metric(weights, 64, "abs_max")
    w_spec = qo.spec_from_case(dict(dtype="sint4", zero=None, groups=((1, 16),), sdt=("f16",)))
    x_spec = qo.spec_from_case(dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",)))
    want = G["search|grid|errors"]
    got = []
    for alpha, beta in pairs:
        s = np.ones(64, np.float32) if alpha == 0 and beta == 0 else so.smooth_scale(a_span, b_span, alpha, beta)
        got.append(so.candidate_error(weights, xs, s, w_spec, x_spec))
    got = np.array(got)
    assert np.all(np.abs(got - want) <= 2e-3 * want), np.abs(got / want - 1).max()
    assert int(np.argmin(got)) == m["best_index"]
    alpha, beta = pairs[m["best_index"]]
    best = so.smooth_scale(a_span, b_span, alpha, beta)
    assert np.all(np.abs(best - G["search|grid|best_scale"]) <= 2 * np.spacing(G["search|grid|best_scale"]))
