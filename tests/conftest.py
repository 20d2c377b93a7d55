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


def row_err(a, b):
    """max over samples of max |a[s]-b[s]| / max |b[s]|: each sample's row judged on its own scale, so a sample with a
    large gradient cannot mask an error in a sample with a small one.  Rows of an all-zero reference are skipped
    (a parameter a sample does not touch, e.g. unused embedding rows) after checking they are zero in `a` too."""
    a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
    b = np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    assert a.shape == b.shape
    scale = np.abs(b).max(1)
    live = scale > 0
    assert not np.abs(a[~live]).any()
    if not live.any():
        return 0.0
    return float((np.abs(a - b).max(1)[live] / scale[live]).max())
