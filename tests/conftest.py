"""pytest configuration: marker registration and shared fixture loaders.

`-m "not gpu"` runs in the build container (no GPU): oracle vs golden vectors, host logic, C-ABI
symbol checks, world_size-2 gloo tests.  `-m gpu` runs on an MI355X: parity through the C-ABI.
"""
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


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_meta(npz):
    return json.loads(str(npz["meta"]))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
