import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uni-encoder-code_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz fixture -> dict of torch tensors (strings/ints stay numpy)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" and a.dtype != np.uint16 else a
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
