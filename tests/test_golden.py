"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).

CPU: the oracle built here, today, still reproduces them bit for bit (pins it across compilers/machines).
GPU: the HIP path reproduces them too (on the GPU box /root/reference and this container do not exist;
the fixtures and the oracle source travel with the repository)."""
import os

import numpy as np
import pytest

from helpers import hip_render, linf, mismatch_report, oracle_render

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(key):
    """The fixtures hold frames of ONE math flavour (rz_oracle_math.h; 1 = the product's default build): a suite that runs on the
    other build of the library (-DRZ_MATH_FLAVOUR=0, RAYZEN_HIP_SO=...) skips them."""
    from oracle import rzo
    from helpers import sync_oracle_flavour
    z = np.load(os.path.join(HERE, "golden", key + ".npz"))
    have = sync_oracle_flavour()
    made = int(z["math_flavour"]) if "math_flavour" in z else 1
    if have != made:
        pytest.skip(f"golden fixtures hold math flavour {made}, the loaded library is flavour {have}")
    return z["accum"], dict(zip([str(n) for n in z["counter_names"]], [int(v) for v in z["counters"]]))


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("key", ["c1_cornell_256x256_4spp_1b", "bunny24_extras_96x54_3spp_5b"])
def test_oracle_reproduces_golden(key):
    mg = _cases()
    c = mg.CASES[key]
    want, wcnt = _load(key)
    got, cnt = oracle_render(mg.make_scene(c["scene"]), c["W"], c["H"], c["spp"], c["bounces"], want_counters=True)
    assert (got.view(np.uint32) == want.view(np.uint32)).all(), mismatch_report(got, want)
    assert {k: cnt[k] for k in wcnt} == wcnt       # (the fixtures hold the ten traversal-side tallies of rounds 1-3; the four shading-side ones of round 4 are compared HIP vs oracle in the parity tests)


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["c1_cornell_256x256_4spp_1b", "bunny24_extras_96x54_3spp_5b"])
def test_hip_reproduces_golden(key):
    mg = _cases()
    c = mg.CASES[key]
    want, wcnt = _load(key)
    got, cnt = hip_render(mg.make_scene(c["scene"]), c["W"], c["H"], c["spp"], c["bounces"], counted=True)
    assert linf(got, want) < 1e-4, mismatch_report(got, want)
    assert (got.view(np.uint32) == want.view(np.uint32)).all(), mismatch_report(got, want)
    assert {k: cnt[k] for k in wcnt} == wcnt       # (the fixtures hold the ten traversal-side tallies of rounds 1-3; the four shading-side ones of round 4 are compared HIP vs oracle in the parity tests)
