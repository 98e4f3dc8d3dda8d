import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference/RayZen"          # present in the build container only, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle and the product libraries once per session (no GPU needed to compile)."""
    from oracle import rzo
    rzo.build()
    from rayzen_amd import build
    build.build_host()
    build.build_hip()       # a no-op when the .so is newer than every source it is built from
    yield


@pytest.fixture(scope="session")
def reference_dir():
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference checkout not present (GPU box)")
    return REFERENCE
