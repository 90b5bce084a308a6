import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """the suite needs libawry_hip.so (git-ignored): build it when it is missing or older than its sources --
    hipcc cross-compiles gfx950 without a GPU -- and only then; the product itself never builds at import time"""
    from awry_amd import build
    if os.path.exists(build.HIPCC) and build.stale():
        build.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi
