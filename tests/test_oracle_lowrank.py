"""Low-rank oracle and host-side configuration against fixtures produced by the reference
(oracle/make_golden.py lowrank).  CPU only.  Tolerances: LAPACK builds differ in the last bits of the singular
vectors, so products are compared to 1e-5 of the weight scale (float32) / one bfloat16 step."""

import json

import numpy as np
import pytest
from conftest import load_golden

import lowrank_oracle as lo
import quant_oracle as qo
from deepcompressor_amd.calib.config import QuantLowRankCalibConfig

G = load_golden("lowrank")
W_SPEC = dict(dtype="sint4", zero=None, groups=((1, 16),), sdt=("f16",))
X_SPEC = dict(dtype="sint8", zero=None, groups=((1, -1),), sdt=("f16",))


@pytest.mark.parametrize("key", sorted(G.meta["dirnames"]))
def test_config_dirnames(key):
    assert QuantLowRankCalibConfig(**json.loads(key)).generate_dirnames(prefix="lr") == G.meta["dirnames"][key]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_oracle_branch(dt):
    w = G[f"branch|{dt}|w"]
    a, b, eff = lo.branch(w, dt, 4)
    tol = 1e-5 if dt == "f32" else 2.0 ** -8
    assert np.abs(eff - G[f"branch|{dt}|effective"]).max() <= tol * np.abs(w).max()
    # the residual of the best rank-4 approximation is the tail of the spectrum
    s = G[f"branch|{dt}|singular"]
    resid = np.linalg.norm(w.astype(np.float64) - eff.astype(np.float64))
    assert abs(resid - np.sqrt((s[4:] ** 2).sum())) <= (1e-5 if dt == "f32" else 2e-2) * resid
    y = (G[f"branch|{dt}|x"] @ a.T) @ b.T
    assert np.abs(y - G[f"branch|{dt}|y"]).max() <= (1e-5 if dt == "f32" else 2.0 ** -6) * np.abs(G[f"branch|{dt}|y"]).max()


@pytest.mark.parametrize("tag,exclusive,compensate", [("shared", False, False), ("exclusive", True, False),
                                                      ("compensate", False, True), ("exclusive_compensate", True, True)])
def test_oracle_joint_quantization(tag, exclusive, compensate):
    ws = [G["joint|w1"], G["joint|w2"]]
    got = lo.joint(ws, "f32", qo.spec_from_case(W_SPEC), 4, exclusive=exclusive, compensate=compensate)
    for i, (q, low) in enumerate(got):
        assert np.abs(low - G[f"joint|{tag}|l{i}"]).max() <= 1e-5 * np.abs(ws[i]).max(), (tag, i)
        # dequantized weights: identical except where a value sat within the SVD noise of a rounding boundary
        want = G[f"joint|{tag}|q{i}"]
        assert (q != want).mean() <= 2e-3, (tag, i, (q != want).mean())


def test_oracle_calibration_errors():
    xs = [G[f"calib|x{i}"] for i in range(3)]
    errors, lows = lo.calibrate_errors(G["calib|w"], xs, qo.spec_from_case(W_SPEC), qo.spec_from_case(X_SPEC), 4, 5)
    want = G["calib|errors"]
    assert np.all(np.abs(errors - want) <= 5e-3 * want), np.abs(errors / want - 1).max()
    assert int(np.argmin(errors)) == G.meta["calib"]["best_index"]
    assert np.abs(lows[G.meta["calib"]["best_index"]] - G["calib|best_effective"]).max() <= 2e-3 * np.abs(G["calib|w"]).max()
