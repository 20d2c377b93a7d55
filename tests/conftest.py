import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Thin reader for tests/golden/*.npz (written by tests/golden/make_golden.py)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"]))
        self.layers = json.loads(str(self.z["layers"])) if "layers" in self.z.files else {}

    def __getitem__(self, k):
        return self.z[k]

    def has(self, k):
        return k in self.z.files

    @property
    def names(self):
        return self.meta["names"]

    def weights(self):
        return {n: self.z["w/" + n] for n in self.names}

    def gs_flat(self):
        B = self.meta["B"]
        return np.concatenate([self.z["gs/" + n].reshape(B, -1) for n in self.names], axis=1)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def rel_err(a, b):
    """max |a-b| / max |b|  (the 1e-5 'relative tolerance' of north_star, per tensor)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)
