import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy access to tests/golden/<name>.npz (+ .json metadata)."""

    def __init__(self, name):
        self.arrays = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
        with open(os.path.join(GOLDEN, f"{name}.json")) as f:
            self.meta = json.load(f)

    def __getitem__(self, key):
        return self.arrays[key]

    def __contains__(self, key):
        return key in self.arrays.files


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope="session")
def golden():
    return load_golden


def bits(a):
    """float32 array -> uint32 bit patterns with -0.0 folded onto +0.0 (the reference never distinguishes them)."""
    a = np.ascontiguousarray(a, dtype=np.float32) + np.float32(0)
    return a.view(np.uint32)


def assert_bit_equal(got, want, what=""):
    got = np.asarray(got, dtype=np.float32).reshape(-1)
    want = np.asarray(want, dtype=np.float32).reshape(-1)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    bad = bits(got) != bits(want)
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} elements differ; first at {i}: {got[i]!r} vs {want[i]!r}")
