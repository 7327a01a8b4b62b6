"""The two-layer random Llama of the stage fixtures, rebuilt from the fixture's parameters."""

import numpy as np
import torch


def bf16_from_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)


def bf16_values(a: np.ndarray) -> np.ndarray:
    """int16 bit patterns of bfloat16 -> float32 values."""
    return (a.astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def state_of(fixture, tag: str) -> dict[str, torch.Tensor]:
    pre = f"{tag}|"
    out = {}
    for k in fixture.arrays.files:
        if k.startswith(pre):
            a = fixture[k]
            out[k[len(pre):]] = bf16_from_bits(a) if a.dtype == np.int16 else torch.from_numpy(a)
    return out


def build(fixture, tag: str = "init", device: str = "cpu"):
    from transformers import LlamaConfig, LlamaForCausalLM

    model = LlamaForCausalLM(LlamaConfig(**fixture.meta["model"], attn_implementation="eager")).to(torch.bfloat16).eval()
    missing, unexpected = model.load_state_dict(state_of(fixture, tag), strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return model.to(device)
