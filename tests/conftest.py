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
    """Lazy view on one tests/golden/*.npz file; keys are '<case>/<name>'."""

    def __init__(self, fname):
        self.z = np.load(os.path.join(GOLDEN, fname))
        self.meta = json.loads(str(self.z["meta"]))

    def cases(self):
        return list(self.meta.keys())

    def get(self, case, name):
        return self.z[f"{case}/{name}"]

    def has(self, case, name):
        return f"{case}/{name}" in self.z.files

    def sub(self, case, prefix):
        """dict of all arrays under '<case>/<prefix>/' keyed by the remainder."""
        p = f"{case}/{prefix}/"
        return {k[len(p):]: self.z[k] for k in self.z.files if k.startswith(p)}


_cache = {}


def golden(fname):
    if fname not in _cache:
        _cache[fname] = Golden(fname)
    return _cache[fname]


@pytest.fixture(scope="session")
def g_cells():
    return golden("cells.npz")


@pytest.fixture(scope="session")
def g_models():
    return golden("models.npz")


@pytest.fixture(scope="session")
def g_integrate():
    return golden("integrate.npz")


@pytest.fixture(scope="session")
def g_training():
    return golden("training.npz")


@pytest.fixture(scope="session")
def g_rambo():
    return golden("rambo.npz")


@pytest.fixture(scope="session")
def g_affine():
    return golden("affine.npz")
