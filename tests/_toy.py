"""The small gated MLP the activation range fixtures were generated on, as a torch module (product side) and as a
numpy forward (oracle side)."""

import numpy as np
import torch
import torch.nn as nn

import quant_oracle as qo


class ToyGated(nn.Module):
    """down(mid(up(x) * sigmoid(gate(x)))); ``mid`` is an identity whose output can be hooked."""

    def __init__(self, channels: int, hidden: int):
        super().__init__()
        self.up = nn.Linear(channels, hidden, bias=False)
        self.gate = nn.Linear(channels, hidden, bias=False)
        self.mid = nn.Identity()
        self.down = nn.Linear(hidden, channels, bias=False)

    def forward(self, x):
        return self.down(self.mid(self.up(x) * torch.sigmoid(self.gate(x))))


def toy_gated_numpy(weights: dict, dtype: str):
    """(mid_of, evaluate): ``mid_of(x)`` = the tensor entering ``down``; ``evaluate(xs, fn_in, fn_mid)`` = outputs per
    batch with ``fn_in`` applied to x and ``fn_mid`` to the hidden tensor.  Every module output is rounded to
    ``dtype`` once (matrix products in float64, so single elements can sit one unit in the last place away from
    torch's float32-accumulating GEMM)."""
    w = {k: v.astype(np.float64) for k, v in weights.items()}
    rnd = lambda a: qo.round_to(dtype, np.asarray(a, dtype=np.float64).astype(np.float32))  # noqa: E731

    def mid_of(x, fn_in=None):
        x = x if fn_in is None else fn_in(x)
        up = rnd(x.astype(np.float64) @ w["up"].T)
        gate = rnd(x.astype(np.float64) @ w["gate"].T)
        sig = rnd(1.0 / (1.0 + np.exp(-gate.astype(np.float64))))
        return rnd(up.astype(np.float64) * sig.astype(np.float64))

    def evaluate(xs, fn_in=None, fn_mid=None):
        outs = []
        for x in xs:
            h = mid_of(x, fn_in)
            h = h if fn_mid is None else fn_mid(h)
            y = rnd(h.astype(np.float64) @ w["down"].T)
            outs.append(y.reshape(-1, y.shape[-1]))
        return outs

    return mid_of, evaluate
