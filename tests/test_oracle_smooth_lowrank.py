"""Smoothing search scored with the low-rank branch in place: the CPU oracle against reference fixtures."""

import numpy as np
import pytest
from conftest import load_golden

import quant_oracle as qo
import smooth_oracle as so
from deepcompressor_amd.calib.config.search import SearchBasedCalibStrategy
from deepcompressor_amd.calib.config.smooth import SmoothCalibConfig, SmoothSpanMode

G = load_golden("smooth_lowrank")
W_SPEC = qo.spec_from_case(dict(dtype="sint4", zero=None, groups=((1, 16),), sdt=("f16",)))
X_SPEC = qo.spec_from_case(dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",)))


def config() -> SmoothCalibConfig:
    return SmoothCalibConfig(spans=[(SmoothSpanMode.AbsMax, SmoothSpanMode.AbsMax)], alpha=0.5, beta=-2, num_grids=6,
                             strategy=SearchBasedCalibStrategy.GridSearch, degree=2, allow_low_rank=True)


@pytest.mark.parametrize("tag", sorted(G.meta))
def test_low_rank_aware_search_matches_reference(tag):
    m = G.meta[tag]
    pairs = config().get_alpha_beta_pairs()
    assert len(pairs) == m["num_candidates"]
    weights = [G[f"{tag}|wa"], G[f"{tag}|wb"]]
    xs = [G[f"{tag}|x{i}"] for i in range(3)]
    a_span = so.channel_metric([x.reshape(-1, 64) for x in xs], 64, "abs_max")
    b_span = so.channel_metric(weights, 64, "abs_max")
    got = []
    for alpha, beta in pairs:
        s = np.ones(64, np.float32) if alpha == 0 and beta == 0 else so.smooth_scale(a_span, b_span, alpha, beta)
        got.append(so.candidate_error_low_rank(weights, xs, s, W_SPEC, X_SPEC, m["low_rank"]["rank"],
                                               exclusive=m["low_rank"].get("exclusive", False), splits=m["splits"]))
    got, want = np.array(got), G[f"{tag}|errors"]
    # float64 SVD of float32 data and float32 GEMMs: errors agree to a fraction of a percent, same winner
    np.testing.assert_allclose(got, want, rtol=1e-2)
    assert int(np.argmin(got)) == m["best_index"]
