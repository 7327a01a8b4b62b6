"""Low-rank branch, joint quantization and alternating calibration on the GPU against the reference's fixtures.

The SVD is LAPACK-on-GPU there and LAPACK-on-CPU in the reference: singular vectors agree up to sign and to the last
bits, so the bar is on products: b @ a within 1e-5 of the weight scale (float32) / one bfloat16 step, dequantized
weights identical except <= 0.2 % of elements that sat within that noise of a rounding boundary, calibration errors
within 5e-3 relative with the same iterate chosen.
"""

import numpy as np
import pytest
import torch
import torch.nn as nn
from _product import to_device, to_np
from conftest import load_golden

from deepcompressor_amd.calib.config import QuantLowRankCalibConfig, SearchBasedCalibObjective
from deepcompressor_amd.calib.lowrank import QuantLowRankCalibrator
from deepcompressor_amd.data.cache import TensorCache, TensorsCache
from deepcompressor_amd.data.dtype import QDType
from deepcompressor_amd.data.utils.reshape import LinearReshapeFn
from deepcompressor_amd.nn.patch import LowRankBranch
from deepcompressor_amd.quantizer import Quantizer, QuantizerConfig
from deepcompressor_amd.quantizer.config.lowrank import QuantLowRankConfig

pytestmark = pytest.mark.gpu
G = load_golden("lowrank")


def _wq(low_rank=None):
    return Quantizer(QuantizerConfig(dtype=QDType.sint4, zero_point=None, group_shapes=((1, 16),), scale_dtypes=(torch.float16,)),
                     key="w", low_rank=low_rank)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_branch_fixture(dt):
    w = to_device(G[f"branch|{dt}|w"], dt)
    lb = LowRankBranch(64, 48, rank=4, weight=w)
    assert lb.a.weight.dtype == w.dtype and lb.a.weight.shape == (4, 64) and lb.b.weight.shape == (48, 4)
    tol = 1e-5 if dt == "f32" else 2.0 ** -8
    eff = to_np(lb.get_effective_weight())
    assert np.abs(eff - G[f"branch|{dt}|effective"]).max() <= tol * np.abs(G[f"branch|{dt}|w"]).max()
    y = to_np(lb(to_device(G[f"branch|{dt}|x"], dt)))
    want = G[f"branch|{dt}|y"]
    assert np.abs(y - want).max() <= (1e-5 if dt == "f32" else 2.0 ** -6) * np.abs(want).max()
    # as a hook the branch adds its output to the module's
    lin = nn.Linear(64, 48, bias=False, device="cuda", dtype=w.dtype)
    x = to_device(G[f"branch|{dt}|x"], dt)
    base = lin(x)
    hook = lb.as_hook().register(lin)
    assert torch.allclose(lin(x).float(), (base + lb(x)).float(), atol=1e-6)
    hook.remove()
    assert torch.equal(lin(x), base)


@pytest.mark.parametrize("tag,kw", [("shared", dict(rank=4)), ("exclusive", dict(rank=4, exclusive=True)),
                                    ("compensate", dict(rank=4, compensate=True)),
                                    ("exclusive_compensate", dict(rank=4, exclusive=True, compensate=True))])
def test_quantize_with_low_rank_fixture(tag, kw):
    ws = [torch.from_numpy(G["joint|w1"]).cuda(), torch.from_numpy(G["joint|w2"]).cuda()]
    qts, lbs = _wq(QuantLowRankConfig(**kw)).quantize_with_low_rank([w.clone() for w in ws])
    assert len(qts) == len(lbs) == 2
    if not kw.get("exclusive") and not kw.get("compensate"):
        assert lbs[0].a is lbs[1].a, "weights sharing their input share the down-projection"
    for i, (qt, lb) in enumerate(zip(qts, lbs)):
        low = to_np(lb.get_effective_weight())
        assert np.abs(low - G[f"joint|{tag}|l{i}"]).max() <= 1e-5 * np.abs(G[f"joint|w{i + 1}"]).max(), (tag, i)
        q, want = to_np(qt.data), G[f"joint|{tag}|q{i}"]
        assert (q != want).mean() <= 2e-3, (tag, i, (q != want).mean())
    qts, lbs = _wq(None).quantize_with_low_rank(ws[0])
    assert lbs is None and len(qts) == 1


def test_calibration_fixture():
    lin = nn.Linear(64, 40, bias=False, device="cuda")
    lin.weight.data = torch.from_numpy(G["calib|w"]).cuda()
    w0 = lin.weight.data.clone()
    xs = [torch.from_numpy(G[f"calib|x{i}"]).cuda() for i in range(3)]
    xq = Quantizer(QuantizerConfig(dtype=QDType.sint8, zero_point=None, group_shapes=((1, -1),), scale_dtypes=(torch.float16,)),
                   channels_dim=-1, key="x")
    cfg = QuantLowRankCalibConfig(rank=4, num_iters=5, early_stop=False, objective=SearchBasedCalibObjective.OutputsError)
    cache = TensorsCache(TensorCache(xs, channels_dim=-1, reshape=LinearReshapeFn(), num_cached=3, num_total=3, num_samples=6,
                                     orig_device=torch.device("cuda")))
    calib = QuantLowRankCalibrator(cfg, _wq(None), xq)
    best = calib.calibrate(x_wgts=[lin.weight], x_mods=[lin], eval_inputs=cache, eval_module=lin, eval_kwargs={})
    assert torch.equal(lin.weight.data, w0) and len(lin._forward_hooks) == 0 and len(lin._forward_pre_hooks) == 0
    errors, want = np.array(calib.error_history), G["calib|errors"]
    assert np.all(np.abs(errors - want) <= 5e-3 * want), np.abs(errors / want - 1).max()
    assert int(np.argmin(errors)) == G.meta["calib"]["best_index"]
    assert np.abs(to_np(best.get_effective_weight()) - G["calib|best_effective"]).max() <= 2e-3 * np.abs(G["calib|w"]).max()
    # early stop ends at the first iteration that does not improve
    cfg2 = QuantLowRankCalibConfig(rank=4, num_iters=50, early_stop=True, objective=SearchBasedCalibObjective.OutputsError)
    calib2 = QuantLowRankCalibrator(cfg2, _wq(None), xq)
    calib2.calibrate(x_wgts=[lin.weight], x_mods=[lin], eval_inputs=cache, eval_module=lin, eval_kwargs={})
    h = calib2.error_history
    assert len(h) < 50 and h[-1] > min(h[:-1]) and all(h[i + 1] <= h[i] for i in range(len(h) - 2))


@pytest.mark.parametrize("shape", [(3072, 3072), (1536, 4608), (4608, 1536)])
def test_leading_triplets_match_full_svd_at_model_size(shape):
    """FLUX-sized weights (hidden 3072; narrower wide / tall cases for the other Gram side): the rank-32 factors from
    the Gram route reproduce the float64 SVD's best rank-32 approximation (Eckart-Young: the residual's Frobenius norm
    equals the tail of the spectrum) and are orthonormal / ordered as an SVD's are."""
    from deepcompressor_amd.nn.patch.lowrank import LowRankBranch, leading_singular_triplets

    g = torch.Generator().manual_seed(shape[0] + shape[1])
    out_f, in_f = shape
    low = torch.randn(out_f, 24, generator=g) @ torch.randn(24, in_f, generator=g) * 0.02
    w = (low + torch.randn(out_f, in_f, generator=g) * 0.02).to(torch.bfloat16).cuda()
    us, vh = leading_singular_triplets(w, 32)
    assert us.dtype == torch.float64 and us.shape == (out_f, 32) and vh.shape == (32, in_f)
    eye = vh @ vh.T
    assert (eye - torch.eye(32, device="cuda", dtype=torch.float64)).abs().max() < 1e-9
    sigma = us.norm(dim=0)
    assert (sigma[:-1] >= sigma[1:]).all()
    s_full = torch.linalg.svdvals(w.double())
    torch.testing.assert_close(sigma, s_full[:32], rtol=1e-9, atol=0)
    resid = (w.double() - us @ vh).norm()
    torch.testing.assert_close(resid, s_full[32:].norm(), rtol=1e-9, atol=0)
    branch = LowRankBranch(in_f, out_f, 32, weight=w)
    eff = branch.get_effective_weight()
    assert eff.dtype == w.dtype and (eff.double() - us @ vh).abs().max() < 2e-2 * float(sigma[0]) / 32


@pytest.mark.parametrize("k", [1, 7, 64, 100, 128])
def test_gram_inverse_factor_matches_float64_lapack(k):
    """The one-launch inverse Cholesky factor of a small Gram matrix (Cholesky QR's dense step) against
    cholesky + solve_triangular in float64: X upper triangular with exact zeros below, X^T G X = I, and the same
    numbers as the LAPACK route to rounding."""
    from deepcompressor_amd.nn.patch.lowrank import _gram_inverse_factor

    g = torch.Generator().manual_seed(k)
    a = torch.randn(3 * k + 5, k, generator=g, dtype=torch.float64) * torch.exp(torch.randn(k, generator=g, dtype=torch.float64))
    gram = (a.T @ a).cuda()
    noisy = gram + 1e-13 * float(gram.abs().max()) * torch.randn(k, k, generator=g, dtype=torch.float64).cuda()  # not symmetric
    x, info = _gram_inverse_factor(noisy.contiguous())
    assert int(info) == 0
    assert torch.equal(x.tril(-1), torch.zeros_like(x))
    sym = (noisy + noisy.T) * 0.5
    want = torch.linalg.solve_triangular(torch.linalg.cholesky(sym, upper=True), torch.eye(k, dtype=torch.float64, device="cuda"),
                                         upper=True)
    scale = want.abs().max()
    assert (x - want).abs().max() <= 1e-9 * scale, float((x - want).abs().max() / scale)
    eye = x.T @ sym @ x
    assert (eye - torch.eye(k, dtype=torch.float64, device="cuda")).abs().max() < 1e-8


def test_gram_inverse_factor_reports_the_first_bad_pivot_and_refuses_large_orders():
    from deepcompressor_amd._lib import DcqError
    from deepcompressor_amd.nn.patch.lowrank import _gram_inverse_factor

    g = torch.Generator().manual_seed(5)
    a = torch.randn(40, 32, generator=g, dtype=torch.float64)
    a[:, 20] = a[:, 3] * 2.0 - a[:, 7]  # column 20 depends on earlier ones: pivot 21 is (numerically) not positive
    gram = (a.T @ a).cuda()
    gram[20, 20] -= 1e-6
    _, info = _gram_inverse_factor(gram)
    assert int(info) == 21, int(info)
    _, ok = _gram_inverse_factor(torch.eye(32, dtype=torch.float64, device="cuda"))
    assert int(ok) == 0
    with pytest.raises(DcqError):
        _gram_inverse_factor(torch.eye(129, dtype=torch.float64, device="cuda"))


BR = load_golden("branch")


@pytest.mark.parametrize("tag", sorted(BR.meta))
def test_branch_checkpoint_round_trip(tag, tmp_path):
    """``branch.pt``: the reference's stored branch (its LowRankBranch.state_dict() for modules that share an input)
    written and read back by the checkpoint writer, then put to work: the shared ``a``, each module's rows of ``b``,
    the effective weight taken out of each module and the hooked outputs, against what the reference's own code
    produced from the same entry; and this package's branch for the same weights has the same keys, shapes, dtypes
    and (up to the sign freedom of singular vectors) the same product."""
    from deepcompressor_amd.app import load_checkpoint, save_checkpoint
    from deepcompressor_amd.nn.patch import add_low_rank_branches

    m = BR.meta[tag]
    dt, rank = m["dtype"], m["rank"]
    state = {k: to_device(BR[f"{tag}|state|{k}"], dt) for k in ("a.weight", "b.weight")}
    written = save_checkpoint(str(tmp_path), branch={"blocks.0.attn.to_q": state})
    assert list(written) == ["branch.pt"]
    back = load_checkpoint(str(tmp_path))["branch.pt"]["blocks.0.attn.to_q"]
    assert sorted(back) == ["a.weight", "b.weight"] and all(v.device.type == "cpu" for v in back.values())
    for k in state:
        assert back[k].dtype == state[k].dtype and torch.equal(back[k], state[k].cpu())
    mods = []
    for i, rows in enumerate(m["outs"]):
        lin = nn.Linear(m["in_features"], rows, bias=False, device="cuda", dtype=state["a.weight"].dtype)
        lin.weight.data = to_device(BR[f"{tag}|w{i}"], dt)
        mods.append(lin)
    joint = torch.cat([lin.weight.data for lin in mods], dim=0)
    branches = add_low_rank_branches(mods, back, rank)
    assert len(branches) == len(mods) and all(b.a is branches[0].a for b in branches)
    x = to_device(BR[f"{tag}|x"], dt)
    step = {"f32": 2e-6, "bf16": 2.0 ** -8, "f16": 2.0 ** -10}[dt]  # one rounding of the weight dtype
    for i, lin in enumerate(mods):
        want = BR[f"{tag}|after{i}"]
        assert np.abs(to_np(lin.weight.data) - want).max() <= step * np.abs(BR[f"{tag}|w{i}"]).max()
        want_y = BR[f"{tag}|y{i}"]
        assert np.abs(to_np(lin(x)) - want_y).max() <= 4 * step * np.abs(want_y).max()
    # this package's own branch for the same joint weight
    own = LowRankBranch(m["in_features"], joint.shape[0], rank=rank, weight=joint).state_dict()
    assert sorted(own) == sorted(state)
    for k in state:
        assert own[k].shape == state[k].shape and own[k].dtype == state[k].dtype
    eff_ref = (state["b.weight"].double() @ state["a.weight"].double())
    eff_own = (own["b.weight"].double() @ own["a.weight"].double())
    assert (eff_own - eff_ref).abs().max() <= 4 * step * joint.abs().max()
