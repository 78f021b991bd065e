import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # LAPACK on 4x4 problems oversubscribes

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 GPU (run on the MI355X box)")


def pkg(name=""):
    return importlib.import_module("ai-camera_amd" + ("." + name if name else ""))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def lib():
    L = pkg("_lib")
    L.load()
    return L


@pytest.fixture(scope="session")
def engines():
    """Seeded engine files (written once under models/; no network for real weights)."""
    ef = pkg("engine_file")
    return ef.ensure_seeded_engines(ROOT)


@pytest.fixture(scope="session")
def gpu(lib):
    if lib.device_count() < 1:
        pytest.skip("no GPU")
    return 0
