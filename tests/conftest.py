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
    build.build_hip()       # a no-op when the library in the tree carries the hash of these very sources and flags
    # the library the suite is about to LOAD must be the one built from this tree: a stale .so beside fresh sources (or
    # an A/B variant left in RAYZEN_HIP_SO) would make every parity claim below a claim about other code
    from rayzen_amd import _lib
    loaded, tree = _lib.hip().rz_source_hash().decode(), build.source_hash()
    if os.environ.get("RAYZEN_HIP_SO"):
        print(f"[conftest] RAYZEN_HIP_SO override: testing {_lib.HIP_SO} (source hash {loaded[:16]}, tree {tree[:16]})", file=sys.stderr)
    else:
        assert loaded == tree, f"{_lib.HIP_SO} was built from other sources ({loaded[:16]}) than this tree ({tree[:16]})"
    from helpers import sync_oracle_flavour
    sync_oracle_flavour()
    yield


@pytest.fixture(scope="session")
def reference_dir():
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference checkout not present (GPU box)")
    return REFERENCE
