"""The CPU oracle's channel reordering (metrics, channel orders, the search) against fixtures captured from the
reference."""

import numpy as np
import pytest
from _toy import toy_gated_numpy
from conftest import load_golden

import quant_oracle as qo
import reorder_oracle as ro

RO = load_golden("reorder")
C = RO.meta["channels"]
XS = [RO[f"x{i}"] for i in range(3)]
WS = [RO[f"w{i}"] for i in range(2)]


@pytest.mark.parametrize("mode", ro.METRICS)
def test_channel_metrics(mode):
    got = ro.channel_metric(XS, WS, mode, C)
    for heads in (1, 4):
        want = RO[f"metric|{mode}|h{heads}"]
        if "AbsMax" in mode and "Product" not in mode:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=4e-7)


def test_abs_normalize_mean():
    got = ro.abs_normalize_mean([x.reshape(-1, C) for x in XS], C, C // 4)
    np.testing.assert_allclose(got, RO["metric|abs_normalize_mean|h4"], rtol=4e-7)


@pytest.mark.parametrize("key", sorted(RO.meta["index_cases"]))
def test_channel_orders(key):
    c = RO.meta["index_cases"][key]
    metric = RO[f"metric|{c['metric']}|h1"]
    got = ro.index_from_metric(metric, c["metric"], c["index_mode"], c["group_size"], c["heads"], c["repeats"])
    np.testing.assert_array_equal(got, RO[key])
    assert sorted(got.tolist()) == list(range(C))


def oracle_run(tag):
    run = RO.meta["runs"][tag]
    wspec, xspec = qo.spec_from_case(run["case_w"]), qo.spec_from_case(run["case_x"])
    manual = ("InputsAbsMax", "Sequential") if run["strategy"] == "Manual" else None
    ws = [WS[0]] if run["module"] == "linear" else [WS[0], (WS[0] * np.float32(0.5) + np.float32(0.01)).astype(np.float32)]
    cands = ro.candidates(XS, ws, C, group_size=run["group_size"], heads=run["heads"], repeats=run["repeats"], manual=manual)

    def outputs(weights, fn_in):
        if run["module"] == "linear":
            return [(x if fn_in is None else fn_in(x)).reshape(-1, C).astype(np.float64) @ weights[0].astype(np.float64).T
                    for x in XS]
        _, evaluate = toy_gated_numpy(dict(up=weights[0], gate=weights[1], down=RO[f"{tag}|w_down"]), "f32")
        return evaluate(XS, fn_in=fn_in)

    base = [np.asarray(y, dtype=np.float32) for y in outputs(ws, None)]

    def evaluate(index):
        wq = [ro.quantize_reordered(w, index, wspec, "f32") for w in ws]
        outs = outputs(wq, lambda x: ro.quantize_reordered(x, index, xspec, "f32"))
        return [np.asarray(y, dtype=np.float32) for y in outs], base

    return run, cands, evaluate


@pytest.mark.parametrize("tag", sorted(RO.meta["runs"]))
def test_candidates_and_search(tag):
    run, cands, evaluate = oracle_run(tag)
    assert len(cands) == run["num_candidates"]
    for i, idx in enumerate(cands):
        want = RO[f"{tag}|cand{i}"]
        if idx is None:
            assert want.size == 0
        else:
            np.testing.assert_array_equal(idx, want)
    best_id, roots = ro.search(cands, evaluate)
    # the fixture holds the summed errors the calibrator was told (before its root)
    np.testing.assert_allclose(np.square(roots), RO[f"{tag}|errors"], rtol=5e-4)
    want = RO[f"{tag}|errors"]
    srt = np.sort(np.sqrt(want))
    if len(srt) == 1 or srt[1] - srt[0] > 1e-3 * srt[0]:  # a clear winner in the reference
        assert best_id == run["best_candidate_id"]
