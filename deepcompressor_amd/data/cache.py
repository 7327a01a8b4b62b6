"""Activation caches handed to the calibration kernels (reference: deepcompressor/data/cache.py:17-281)."""

from __future__ import annotations

import math
import typing as tp
from collections import OrderedDict
from dataclasses import dataclass, field

import torch

from .utils.reshape import ReshapeFn

__all__ = ["ModuleForwardInput", "TensorCache", "TensorsCache", "IOTensorsCache"]


def _map_tensors(fn, obj):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(fn, v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(fn, v) for v in obj)
    return obj


@dataclass
class ModuleForwardInput:
    args: list[tp.Any] = field(default_factory=list)
    kwargs: dict[str, tp.Any] = field(default_factory=dict)

    def to(self, device) -> "ModuleForwardInput":
        move = lambda t: t.to(device=device)  # noqa: E731
        return ModuleForwardInput(args=_map_tensors(move, self.args), kwargs=_map_tensors(move, self.kwargs))

    def update(self, x: dict[str | int, tp.Any] | None = None) -> "ModuleForwardInput":
        args, kwargs = list(self.args), dict(self.kwargs)
        for k, v in (x or {}).items():
            if isinstance(k, int):
                args[k] = v
            else:
                kwargs[k] = v
        return ModuleForwardInput(args=args, kwargs=kwargs)


@dataclass
class TensorCache:
    """Cached activations of one tensor: a list of batches, the channel dimension and a 2-D reshape."""

    data: list[torch.Tensor] = field(default_factory=list)
    channels_dim: int = 1
    reshape: ReshapeFn = ReshapeFn()
    num_cached: int = 0
    num_total: int = 0
    num_samples: int = 0
    orig_device: torch.device | str = torch.device("cpu")

    def clear(self):
        self.data.clear()
        self.num_cached = 0

    def get_factory_kwargs(self, **kwargs) -> dict[str, tp.Any]:
        for name in ("channels_dim", "reshape", "num_cached", "num_total", "orig_device"):
            kwargs.setdefault(name, getattr(self, name))
        return kwargs

    def get_standardized_data(self, reshape: bool = False) -> list[torch.Tensor]:
        """Batches with the dims ahead of ``channels_dim`` flattened (and optionally reshaped to 2-D)."""
        flat = [x.view(-1, *x.shape[self.channels_dim :]) for x in self.data]
        return [self.reshape(x) for x in flat] if reshape else flat

    def repartition(self, max_batch_size: int, max_size: int, standardize: bool, reshape: bool) -> "TensorCache":
        assert self.data, "No data to relocate."
        assert max_batch_size != 0 and max_size != 0
        assert all(x.shape == self.data[0].shape for x in self.data), "All tensors must have the same shape."
        data, dim, fn = self.data, self.channels_dim, self.reshape
        if standardize:
            data, dim = [x.view(-1, *x.shape[dim:]) for x in data], 1
            if reshape:
                data, dim, fn = [fn(x) for x in data], -1, ReshapeFn()
        dim %= data[0].ndim
        before = data[0].shape[0] * len(data)
        if max_batch_size > 0:
            rows = data[0].shape[0]
            if rows > max_batch_size:
                data = [x[i * max_batch_size : (i + 1) * max_batch_size] for x in data
                        for i in range(rows // max_batch_size)]
            rows = data[0].shape[0]
            if max_size > 0 and rows * len(data) > max_size:
                assert max_size >= rows, "max_size must be greater than or equal to batch_size."
                data = data[:: len(data) // (max_size // rows)]
        else:
            assert max_size < 0, "max_size must be negative if max_batch_size is negative."
        kept = data[0].shape[0] * len(data) / before
        return TensorCache(data, channels_dim=dim, reshape=fn, orig_device=self.orig_device,
                           num_cached=int(math.ceil(kept * self.num_cached)),
                           num_total=int(math.ceil(kept * self.num_total)),
                           num_samples=int(math.ceil(kept * self.num_samples)))


class TensorsCache:
    """Ordered mapping argument-position / keyword -> TensorCache."""

    def __init__(self, tensors: "OrderedDict[str | int, TensorCache] | TensorCache") -> None:
        self.tensors = OrderedDict({0: tensors}) if isinstance(tensors, TensorCache) else tensors

    num_tensors = property(lambda self: len(self.tensors))

    def front(self) -> TensorCache:
        return next(iter(self.tensors.values()))

    def items(self):
        return self.tensors.items()

    def keys(self):
        return self.tensors.keys()

    def values(self):
        return self.tensors.values()

    def __getitem__(self, key) -> TensorCache:
        return self.tensors[key]

    def __iter__(self):
        return iter(self.tensors.values())

    def __len__(self) -> int:
        return len(self.tensors)

    def clear(self):
        for t in self.tensors.values():
            t.clear()

    def set_num_samples(self, num_samples: int):
        for t in self.tensors.values():
            t.num_samples = num_samples

    def extract(self, index: int, kwargs: dict[str, tp.Any]) -> ModuleForwardInput:
        args, kw = [], dict(kwargs)
        for key, t in self.tensors.items():
            value = t.data[index].to(t.orig_device, non_blocking=True)
            if isinstance(key, int):
                assert len(args) == key, f"Expected {key} args, but got {len(args)}"
                args.append(value)
            else:
                kw[key] = value
        return ModuleForwardInput(args=args, kwargs=kw)


class IOTensorsCache:
    def __init__(self, inputs=None, outputs=None):
        self.inputs = TensorsCache(inputs) if isinstance(inputs, TensorCache) else inputs
        self.outputs = TensorsCache(outputs) if isinstance(outputs, TensorCache) else outputs

    def clear(self):
        for c in (self.inputs, self.outputs):
            if c is not None:
                c.clear()

    def set_num_samples(self, num_samples: int):
        for c in (self.inputs, self.outputs):
            if c is not None:
                c.set_num_samples(num_samples)
