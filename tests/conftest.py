"""pytest configuration: the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` runs here without a GPU (oracle vs golden vectors, host logic, C-ABI
symbol checks); ``-m gpu`` runs on a real MI355X and is the parity suite proper: HIP path
(through the C ABI) vs the oracle and the committed golden fixtures.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _device_count():
    try:
        from pyamg_amd import _capi
        return _capi.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    if _device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def kernels_npz():
    return np.load(GOLDEN / "kernels.npz")


def golden_hierarchies():
    return sorted(p.stem[len("hier_"):] for p in GOLDEN.glob("hier_*.npz"))


@pytest.fixture(scope="session")
def load_hier():
    from pyamg_amd.hierarchy import load_spec
    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = load_spec(GOLDEN / f"hier_{name}.npz")
        return cache[name]
    return _load
