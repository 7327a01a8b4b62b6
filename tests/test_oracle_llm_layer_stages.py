"""The rotation / smoothing stage oracle (a composition of the rotation and smoothing oracles) and the model structs
against fixtures captured from the reference's ``rotate_llm`` / ``smooth_llm_layer`` on a two-layer random Llama."""

import numpy as np
import pytest
from _tiny_llama import bf16_values, build
from conftest import load_golden

import llm_stage_oracle as lo
import smooth_oracle as so

LS = load_golden("llm_layer_stages")
MODEL = LS.meta["model"]
HEADS, KV = MODEL["num_attention_heads"], MODEL["num_key_value_heads"]
HC = MODEL["hidden_size"] // HEADS


def values(tag):
    pre = f"{tag}|"
    return {k[len(pre):]: bf16_values(LS[k]) if LS[k].dtype == np.int16 else LS[k] for k in LS.arrays.files
            if k.startswith(pre)}


def ulp_bf16(got, want):
    return np.abs(got.astype(np.float64) - want) / np.maximum(np.spacing(np.abs(want).astype(np.float32)) * 65536, 1e-45)


def test_rotation_stage_oracle_matches_reference():
    got = lo.rotate_llm(values("init"), LS["rotation"], MODEL["num_hidden_layers"], HC)
    want = values("rotated")
    assert sorted(got) == sorted(want)
    for k in want:
        u = ulp_bf16(got[k], want[k])
        # float64 products summed in another order can land on the other side of a rounding edge
        assert u.max() <= 1 and (u == 0).mean() >= 0.999, (k, u.max(), (u == 0).mean())


def test_rotation_keeps_the_logits():
    a, b = LS["logits|init"], LS["logits|rotated"]
    assert np.abs(a - b).max() <= 0.03 * np.abs(a).max()


def test_manual_smoothing_stage_oracle_matches_reference():
    """The five smoothing steps in the reference's order (each sees the weights the earlier ones left): the scale
    the oracle finds is within 4 ulp of the reference's (a quotient of two float32 powers); with the reference's
    scale applied, the parameters after the stage are exact."""
    state, want = values("rotated"), values("smooth|manual|state")
    acts = lambda name, side="inputs": list(bf16_values(LS[f"acts|{name}|{side}"]))  # noqa: E731
    p = "model.layers.0"
    up, down, repeats = so.upscale, so.downscale, HEADS // KV
    w = lambda n: state[f"{p}.{n}.weight"]  # noqa: E731

    def close(got, key):
        ref = LS[f"smooth|manual|scale|{key}"]
        assert (np.abs(got.astype(np.float64) - ref) <= 4 * np.spacing(ref)).all(), key
        return ref

    k_span, q_span = so.attention_spans(acts(f"{p}.self_attn.q_proj", "outputs"), acts(f"{p}.self_attn.k_proj", "outputs"),
                                        num_heads=HEADS, num_head_repeats=repeats, with_rope=True, key_mode="AbsMax",
                                        query_mode="AbsMax")
    s = close(so.smooth_scale(k_span, q_span, 0.5, 0), f"{p}.self_attn.attn_k")
    state[f"{p}.self_attn.q_proj.weight"] = up(w("self_attn.q_proj"), "bf16", s, 0)
    state[f"{p}.self_attn.k_proj.weight"] = down(w("self_attn.k_proj"), "bf16", so.downscale_for_producer(s, HEADS, repeats), 0)
    qkv = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")
    s = close(lo.manual_proj_scale(acts(f"{p}.self_attn.q_proj"), [w(n) for n in qkv], 0.3, 0.7), f"{p}.self_attn.v_proj")
    for n in qkv:
        state[f"{p}.{n}.weight"] = up(w(n), "bf16", s, 1)
    state[f"{p}.input_layernorm.weight"] = down(w("input_layernorm"), "bf16", s, 0)
    s = close(lo.manual_proj_scale(acts(f"{p}.self_attn.o_proj"), [w("self_attn.o_proj")], 0.3, 0.7, HEADS, repeats),
              f"{p}.self_attn.o_proj")
    state[f"{p}.self_attn.o_proj.weight"] = up(w("self_attn.o_proj"), "bf16", s, 1)
    state[f"{p}.self_attn.v_proj.weight"] = down(w("self_attn.v_proj"), "bf16", so.downscale_for_producer(s, HEADS, repeats), 0)
    s = close(lo.manual_proj_scale(acts(f"{p}.mlp"), [w("mlp.up_proj"), w("mlp.gate_proj")], 0.3, 0.7), f"{p}.mlp")
    for n in ("mlp.up_proj", "mlp.gate_proj"):
        state[f"{p}.{n}.weight"] = up(w(n), "bf16", s, 1)
    state[f"{p}.post_attention_layernorm.weight"] = down(w("post_attention_layernorm"), "bf16", s, 0)
    s = close(lo.manual_proj_scale(acts(f"{p}.mlp.down_proj"), [w("mlp.down_proj")], 0.3, 0.7), f"{p}.mlp.down_proj")
    state[f"{p}.mlp.down_proj.weight"] = up(w("mlp.down_proj"), "bf16", s, 1)
    state[f"{p}.mlp.up_proj.weight"] = down(w("mlp.up_proj"), "bf16", s, 0)
    for k in want:
        np.testing.assert_array_equal(state[k], want[k], err_msg=k)


def test_structs_name_modules_like_the_reference():
    pytest.importorskip("transformers")
    from deepcompressor_amd.app import LlmModelStruct

    struct = LlmModelStruct.construct(build(LS))
    layer = struct.backbone_struct.layer_structs[0]
    attn, ffn = layer.attn_struct, layer.ffn_struct
    watched = [attn.name, attn.q_proj_name, attn.k_proj_name, attn.o_proj_name, ffn.name, ffn.down_proj_names[0],
               attn.v_proj_name, *ffn.up_proj_names]
    assert watched == LS.meta["watched"]
    assert (attn.qkv_proj_key, attn.out_proj_key, ffn.up_proj_key, ffn.down_proj_key, attn.q_key, attn.k_key) == (
        "qkv_proj", "out_proj", "up_proj", "down_proj", "attn_q", "attn_k")
    assert sorted([f"{attn.name}.{attn.k_rkey}", attn.v_proj_name, attn.o_proj_name, ffn.name, ffn.down_proj_names[0]]
                  ) == LS.meta["smooth|manual"]["scales"]
    assert ffn.up_proj_names == ["model.layers.0.mlp.up_proj", "model.layers.0.mlp.gate_proj"]
    assert attn.config.num_query_heads == HEADS and attn.config.num_head_repeats == HEADS // KV
    assert struct.num_head_channels == HC and struct.backbone_struct.num_channels == MODEL["hidden_size"]
    assert [name for _, name, *_ in layer.named_key_modules()] == LS.meta["chain"]["weight_ranges_in_order"]


def v_proj_order(index: np.ndarray) -> np.ndarray:
    """reorder.py:138-149: the order inside the first query head of every group, as an order of v_proj's rows."""
    n = index.size
    idx = index.reshape(HEADS, HC) - np.arange(0, n, HC).reshape(HEADS, 1)
    idx = idx.reshape(KV, HEADS // KV, HC)[:, 0, :] + np.arange(0, n // (HEADS // KV), HC).reshape(KV, 1)
    return idx.reshape(-1)


def assert_same_sort(metric: np.ndarray, got: np.ndarray, want: np.ndarray) -> None:
    assert sorted(got.tolist()) == sorted(want.tolist()) == list(range(metric.size))
    np.testing.assert_array_equal(metric[got], metric[want])
    values_, counts = np.unique(metric, return_counts=True)
    unique = np.isin(metric[want], values_[counts == 1])
    np.testing.assert_array_equal(got[unique], want[unique])


def test_manual_reorder_stage_oracle_matches_reference():
    import reorder_oracle as rr

    state, want = values("rotated"), values("reorder|manual|state")
    acts = lambda name: list(bf16_values(LS[f"acts|{name}|inputs"]))  # noqa: E731
    p = "model.layers.0"
    assert LS.meta["reorder|manual"]["indexes"] == sorted([f"{p}.self_attn.o_proj", f"{p}.mlp.down_proj"])
    o_w = state[f"{p}.self_attn.o_proj.weight"]
    (o_index,) = rr.candidates(acts(f"{p}.self_attn.o_proj"), [o_w], o_w.shape[1], 16, HEADS, HEADS // KV,
                               manual=("InputsAbsMax", "Sequential"))
    np.testing.assert_array_equal(o_index, LS[f"reorder|manual|index|{p}.self_attn.o_proj"])
    d_w = state[f"{p}.mlp.down_proj.weight"]
    (d_index,) = rr.candidates(acts(f"{p}.mlp.down_proj"), [d_w], d_w.shape[1], 16, manual=("InputsAbsMax", "Sequential"))
    # bfloat16 activations tie in their abs-max; the reference's argsort is not stable, so tied channels may swap:
    # both orders must sort the metric identically and agree wherever the metric value is unique
    d_ref = LS[f"reorder|manual|index|{p}.mlp.down_proj"]
    assert_same_sort(rr.channel_metric(acts(f"{p}.mlp.down_proj"), [d_w], "InputsAbsMax", d_w.shape[1]), d_index, d_ref)
    d_index = d_ref
    state[f"{p}.self_attn.o_proj.weight"] = o_w[:, o_index]
    state[f"{p}.self_attn.v_proj.weight"] = state[f"{p}.self_attn.v_proj.weight"][v_proj_order(o_index)]
    state[f"{p}.mlp.down_proj.weight"] = d_w[:, d_index]
    for n in ("up_proj", "gate_proj"):
        state[f"{p}.mlp.{n}.weight"] = state[f"{p}.mlp.{n}.weight"][d_index]
    for k in want:
        np.testing.assert_array_equal(state[k], want[k], err_msg=k)
