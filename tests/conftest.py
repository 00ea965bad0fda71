"""pytest configuration: the ``gpu`` marker and import paths.

``-m "not gpu"`` runs on the CPU-only build container (oracle vs golden fixtures, host
logic, C-ABI symbol check, gloo world_size-2).  ``-m gpu`` needs an MI355X and calls the
HIP library through its C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


@pytest.fixture(autouse=True)
def _seed_torch():
    """Every test starts from the same torch generator state: module constructors (Glow, nn.Linear) and the
    default RNVP masks draw from it, and a test must not depend on which tests ran before it."""
    import torch

    torch.manual_seed(1234)
    yield
