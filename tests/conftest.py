import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (checker). Built on demand with gcc."""
    from ar_voxel_project_amd import build
    build.build_oracle()
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def arvx():
    """libarvx.so through ctypes. The prebuilt .so travels with the repo; build
    it here only if it is missing (hipcc cross-compiles without a GPU)."""
    from ar_voxel_project_amd import build, capi
    if not os.path.exists(capi.LIB_PATH):
        build.build_library()
    capi.load_library()
    return capi
