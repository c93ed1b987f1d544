import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Safety net: (re)build the in-tree shared libraries when they are missing (hipcc and gcc are present
    in this image, on the build container and on the GPU box alike)."""
    from radarslampy_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        from radarslampy_amd import build as hipbuild
        hipbuild.build()
    import oracle
    oracle.build()
