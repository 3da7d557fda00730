import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu tests are skipped (not failed) when no device is visible, so `-m "not gpu"` and a plain run both pass on CPU
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """npz fixture with '/'-separated keys; tensors come back as torch tensors."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __contains__(self, k):
        return k in self.z.files

    def np(self, k):
        return self.z[k]

    def __getitem__(self, k):
        a = self.z[k]
        if a.dtype.kind in "US":
            return a
        return torch.from_numpy(np.array(a))

    def sub(self, prefix):
        return {k[len(prefix):]: self[k] for k in self.z.files if k.startswith(prefix)}

    def keys(self):
        return list(self.z.files)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get
