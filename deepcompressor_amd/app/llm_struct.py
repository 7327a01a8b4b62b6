"""A decoder-only language model described by the modules the calibration stages touch
(reference: deepcompressor/app/llm/nn/struct.py:225-947, nn/struct/attn.py:140-480).

Attribute names follow the reference's struct classes, so the stage drivers in ``llm_stages`` read like the
reference's and also accept the reference's own struct objects.  ``LlmModelStruct.construct`` recognises the
Llama family layout (``model.embed_tokens``, ``model.layers[i].{input_layernorm, self_attn.{q,k,v,o}_proj,
post_attention_layernorm, mlp.{up,gate,down}_proj}``, ``model.norm``, ``lm_head``) by attribute names, without
importing the model library; other layouts can fill the dataclasses by hand.
"""

from __future__ import annotations

import typing as tp
from dataclasses import dataclass, field

import torch.nn as nn

__all__ = ["AttentionConfigStruct", "FeedForwardConfigStruct", "LlmSelfAttentionStruct", "LlmFeedForwardStruct",
           "LlmTransformerBlockStruct", "LlmTransformerStruct", "LlmModelStruct"]


def _join(*parts: str) -> str:
    return ".".join(p for p in parts if p)


@dataclass
class AttentionConfigStruct:
    hidden_size: int
    inner_size: int
    num_query_heads: int
    num_key_value_heads: int
    with_qk_norm: bool = False
    with_rope: bool = True
    linear_attn: bool = False

    @property
    def num_head_channels(self) -> int:
        return self.inner_size // self.num_query_heads

    @property
    def num_head_repeats(self) -> int:
        return self.num_query_heads // self.num_key_value_heads


@dataclass
class FeedForwardConfigStruct:
    hidden_size: int
    intermediate_size: int
    num_experts: int = 1


@dataclass
class LlmSelfAttentionStruct:
    """struct.py:233-340: the attention block, its four projections, and the modules whose outputs are the queries /
    keys / values entering the attention product (``q``, ``k``, ``v``)."""

    module: nn.Module
    name: str
    config: AttentionConfigStruct
    q_proj: nn.Linear
    k_proj: nn.Linear
    v_proj: nn.Linear
    o_proj: nn.Linear
    q: nn.Module
    k: nn.Module
    v: nn.Module
    q_proj_rname: str = "q_proj"
    k_proj_rname: str = "k_proj"
    v_proj_rname: str = "v_proj"
    o_proj_rname: str = "o_proj"
    q_rname: str = "q_proj"
    k_rname: str = "k_proj"
    v_rname: str = "v_proj"
    kwargs: tuple[str, ...] = ("attention_mask", "position_ids", "past_key_value", "output_attentions", "use_cache",
                               "position_embeddings", "cache_position")
    parent: tp.Optional["LlmTransformerBlockStruct"] = field(default=None, repr=False)
    key: str = ""

    qkv_proj_rkey: tp.ClassVar[str] = "qkv_proj"
    out_proj_rkey: tp.ClassVar[str] = "out_proj"
    q_rkey: tp.ClassVar[str] = "attn_q"
    k_rkey: tp.ClassVar[str] = "attn_k"
    v_rkey: tp.ClassVar[str] = "attn_v"

    def __post_init__(self) -> None:
        self.qkv_proj = [self.q_proj, self.k_proj, self.v_proj]
        self.out_proj = self.o_proj
        for part in ("q_proj", "k_proj", "v_proj", "o_proj", "q", "k", "v"):
            setattr(self, f"{part}_name", _join(self.name, getattr(self, f"{part}_rname")))
        self.qkv_proj_names = [self.q_proj_name, self.k_proj_name, self.v_proj_name]
        self.qkv_proj_rnames = [self.q_proj_rname, self.k_proj_rname, self.v_proj_rname]
        self.out_proj_name, self.out_proj_rname = self.o_proj_name, self.o_proj_rname
        keyed = lambda rkey: f"{self.key}_{rkey}" if self.key else rkey  # noqa: E731
        self.qkv_proj_key, self.out_proj_key = keyed(self.qkv_proj_rkey), keyed(self.out_proj_rkey)
        self.q_key, self.k_key, self.v_key = keyed(self.q_rkey), keyed(self.k_rkey), keyed(self.v_rkey)

    def filter_kwargs(self, kwargs: dict) -> dict:
        """The layer's keyword arguments that the attention block takes."""
        return {k: v for k, v in kwargs.items() if k in self.kwargs}

    def named_key_modules(self):
        """(key, name, linear, owner struct, field) of the projections, in the reference's order
        (nn/struct/attn.py:383-397)."""
        yield self.qkv_proj_key, self.q_proj_name, self.q_proj, self, "q_proj"
        yield self.qkv_proj_key, self.k_proj_name, self.k_proj, self, "k_proj"
        yield self.qkv_proj_key, self.v_proj_name, self.v_proj, self, "v_proj"
        yield self.out_proj_key, self.o_proj_name, self.o_proj, self, "o_proj"


@dataclass
class LlmFeedForwardStruct:
    """struct.py:343-450: up projections first (``up_proj`` then ``gate_proj``), one down projection per expert."""

    module: nn.Module
    name: str
    config: FeedForwardConfigStruct
    up_projs: list[nn.Linear]
    down_projs: list[nn.Linear]
    up_proj_rnames: list[str] = field(default_factory=lambda: ["up_proj", "gate_proj"])
    down_proj_rnames: list[str] = field(default_factory=lambda: ["down_proj"])
    moe_gate: nn.Linear | None = None
    parent: tp.Optional["LlmTransformerBlockStruct"] = field(default=None, repr=False)
    key: str = ""

    up_proj_rkey: tp.ClassVar[str] = "up_proj"
    down_proj_rkey: tp.ClassVar[str] = "down_proj"
    moe_gate_rkey: tp.ClassVar[str] = "moe_gate"

    def __post_init__(self) -> None:
        assert self.config.num_experts == 1, "mixture-of-experts blocks are filled in by hand"
        self.experts, self.expert_names = [self.module], [self.name]
        self.up_proj_names = [_join(self.name, r) for r in self.up_proj_rnames]
        self.down_proj_names = [_join(self.name, r) for r in self.down_proj_rnames]
        self.moe_gate_name = _join(self.name, "gate") if self.moe_gate is not None else ""
        keyed = lambda rkey: f"{self.key}_{rkey}" if self.key else rkey  # noqa: E731
        self.up_proj_key, self.down_proj_key = keyed(self.up_proj_rkey), keyed(self.down_proj_rkey)
        self.moe_gate_key = keyed(self.moe_gate_rkey)

    def named_key_modules(self):
        """nn/struct/attn.py:561-573."""
        for name, module in zip(self.up_proj_names, self.up_projs, strict=True):
            yield self.up_proj_key, name, module, self, "up_proj"
        for name, module in zip(self.down_proj_names, self.down_projs, strict=True):
            yield self.down_proj_key, name, module, self, "down_proj"


@dataclass
class LlmTransformerBlockStruct:
    """struct.py:453-612: one decoder layer."""

    module: nn.Module
    name: str
    pre_attn_norm: nn.Module | None
    pre_attn_norm_rname: str
    attn_struct: LlmSelfAttentionStruct
    pre_ffn_norm: nn.Module | None
    pre_ffn_norm_rname: str
    ffn_struct: LlmFeedForwardStruct
    post_attn_norms: list[nn.Module] = field(default_factory=list)
    post_ffn_norm: nn.Module | None = None

    def __post_init__(self) -> None:
        self.attn_struct.parent = self.ffn_struct.parent = self
        self.attn_structs = [self.attn_struct]

    def named_key_modules(self):
        yield from self.attn_struct.named_key_modules()
        yield from self.ffn_struct.named_key_modules()

    @classmethod
    def construct(cls, module: nn.Module, name: str) -> "LlmTransformerBlockStruct":
        attn, mlp = module.self_attn, module.mlp
        conf = getattr(attn, "config", None)
        q_out, hidden = attn.q_proj.weight.shape
        head_dim = getattr(attn, "head_dim", None) or (hidden // conf.num_attention_heads)
        num_q, num_kv = q_out // head_dim, attn.k_proj.weight.shape[0] // head_dim
        rotary = hasattr(attn, "q_rotary_emb")
        attn_struct = LlmSelfAttentionStruct(
            module=attn, name=_join(name, "self_attn"),
            config=AttentionConfigStruct(hidden_size=hidden, inner_size=q_out, num_query_heads=num_q,
                                         num_key_value_heads=num_kv),
            q_proj=attn.q_proj, k_proj=attn.k_proj, v_proj=attn.v_proj, o_proj=attn.o_proj,
            q=attn.q_rotary_emb if rotary else attn.q_proj, k=attn.k_rotary_emb if rotary else attn.k_proj,
            v=attn.v_proj, q_rname="q_rotary_emb" if rotary else "q_proj",
            k_rname="k_rotary_emb" if rotary else "k_proj")
        ffn_struct = LlmFeedForwardStruct(
            module=mlp, name=_join(name, "mlp"),
            config=FeedForwardConfigStruct(hidden_size=hidden, intermediate_size=mlp.up_proj.weight.shape[0]),
            up_projs=[mlp.up_proj, mlp.gate_proj], down_projs=[mlp.down_proj])
        return cls(module=module, name=name, pre_attn_norm=module.input_layernorm,
                   pre_attn_norm_rname="input_layernorm", attn_struct=attn_struct,
                   pre_ffn_norm=module.post_attention_layernorm, pre_ffn_norm_rname="post_attention_layernorm",
                   ffn_struct=ffn_struct)


@dataclass
class LlmTransformerStruct:
    """struct.py:615-778: embeddings, decoder layers, final norm."""

    module: nn.Module
    name: str
    embed_tokens: nn.Embedding
    layer_structs: list[LlmTransformerBlockStruct]
    norm_out: nn.Module | None
    norm_out_rname: str = "norm"
    embed_positions: nn.Embedding | None = None
    norm_in: nn.Module | None = None
    proj_in: nn.Linear | None = None
    proj_out: nn.Linear | None = None

    def __post_init__(self) -> None:
        self.embed_tokens_name = _join(self.name, "embed_tokens")
        self.norm_out_name = _join(self.name, self.norm_out_rname)
        self.num_channels = self.embed_tokens.weight.shape[1]


@dataclass
class LlmModelStruct:
    """struct.py:781-947: the backbone and the language-model head."""

    module: nn.Module
    backbone_struct: LlmTransformerStruct
    head: nn.Linear | None
    head_name: str = "lm_head"

    @property
    def num_head_channels(self) -> int:
        return self.backbone_struct.layer_structs[0].attn_struct.config.num_head_channels

    def iter_transformer_block_structs(self) -> tp.Iterator[LlmTransformerBlockStruct]:
        yield from self.backbone_struct.layer_structs

    @classmethod
    def construct(cls, model: nn.Module) -> "LlmModelStruct":
        backbone = model.model
        layers = [LlmTransformerBlockStruct.construct(layer, f"model.layers.{i}")
                  for i, layer in enumerate(backbone.layers)]
        return cls(module=model, backbone_struct=LlmTransformerStruct(
            module=backbone, name="model", embed_tokens=backbone.embed_tokens, layer_structs=layers,
            norm_out=backbone.norm), head=getattr(model, "lm_head", None))
