"""The built-ins of oracle/rz_oracle_math.h: accuracy (they are meant to be sin/cos/acos, not just deterministic), known
answers, and the GLSL helper semantics.  Flavour 0 = rounds 1-4's binary64 definitions (correctly rounded); flavour 1 = Mesa
llvmpipe's, the default since round 5 -- pinned bit for bit against llvmpipe's own tables in tests/test_glref.py."""
import math

import numpy as np
import pytest

from oracle import rzo


def _ulp_err(got, want):
    got = np.float32(got)
    want32 = np.float32(want)
    if got == want32:
        return 0.0
    return abs(float(got) - want) / float(np.spacing(np.abs(want32)) or 1e-45)


@pytest.fixture
def flavour0():
    with rzo.math_flavour(0):
        yield


@pytest.fixture
def flavour1():
    with rzo.math_flavour(1):
        yield


def test_sin_cos_match_correctly_rounded_libm_over_the_rng_argument_range(flavour0):
    L = rzo.lib()
    rng = np.random.default_rng(1234)
    xs = np.concatenate([rng.uniform(-8, 8, 3000), rng.uniform(-3e5, 3e5, 3000), rng.uniform(-4e10, 4e10, 3000),
                         [0.0, 1e-30, -1e-30, math.pi, 1.5707963705062866, 43758.5453, 78.233]]).astype(np.float32)
    worst = 0.0
    exact = 0
    for x in xs:
        xf = float(x)
        s, c = L.rzo_sin_f(xf), L.rzo_cos_f(xf)
        worst = max(worst, _ulp_err(s, math.sin(xf)), _ulp_err(c, math.cos(xf)))
        exact += int(np.float32(s) == np.float32(math.sin(xf))) + int(np.float32(c) == np.float32(math.cos(xf)))
    assert worst <= 1.0, worst                       # never worse than 1 ulp ...
    assert exact >= 2 * len(xs) - 2                  # ... and correctly rounded essentially always


def test_acos_accuracy_and_endpoints(flavour0):
    L = rzo.lib()
    rng = np.random.default_rng(5)
    for x in np.concatenate([rng.uniform(-1, 1, 4000), [0.0, 0.5, -0.5, 0.4999999, 0.99999994]]).astype(np.float32):
        assert _ulp_err(L.rzo_acos_f(float(x)), math.acos(float(x))) <= 1.0
    assert L.rzo_acos_f(1.0) == 0.0
    assert np.float32(L.rzo_acos_f(-1.0)) == np.float32(math.pi)
    assert math.isnan(L.rzo_acos_f(float("nan")))


@pytest.mark.parametrize("flavour", [0, 1])
def test_rand_is_fract_sin_dot(flavour):
    """FS:188-190 with the flavour's sin: recompute in numpy float32, operation by operation."""
    L = rzo.lib()
    rng = np.random.default_rng(2)
    with rzo.math_flavour(flavour):         # (restores the process default -- the loaded product library's -- afterwards)
        for _ in range(500):
            x, y = np.float32(rng.uniform(0, 3000)), np.float32(rng.uniform(0, 3000))
            d = np.float32(np.float32(x * np.float32(12.9898)) + np.float32(y * np.float32(78.233)))
            s = np.float32(L.rzo_sin_f(float(d)))
            p = np.float32(s * np.float32(43758.5453))
            want = np.float32(p - np.floor(p))
            got = np.float32(L.rzo_rand_f(float(x), float(y)))
            assert got == want and 0.0 <= got < 1.0


def test_rand_known_answers(flavour0):
    """Values committed from this oracle (regression pin; the reference publishes none)."""
    L = rzo.lib()
    assert L.rzo_rand_f(0.3, 0.7) == 0.08203125
    assert L.rzo_rand_f(1234.5, 6789.1) == 0.755859375
    assert L.rzo_rand_f(0.0, 0.0) == 0.0


def test_llvmpipe_flavour_is_an_accurate_sine_where_sines_are_expected(flavour1):
    """Flavour 1 is Cephes' single-precision routine: within an ulp or two of the true value for |x| < 1e4 (the camera jitter's
    arguments), and still a deterministic function -- if no longer a sine -- at the hash's 1e8 ... 1e11."""
    L = rzo.lib()
    rng = np.random.default_rng(77)
    for x in np.concatenate([rng.uniform(-8, 8, 2000), rng.uniform(-1e4, 1e4, 2000)]).astype(np.float32):
        assert abs(L.rzo_sin_f(float(x)) - math.sin(float(x))) <= 1.3e-7
        assert abs(L.rzo_cos_f(float(x)) - math.cos(float(x))) <= 1.3e-7
    for x in rng.uniform(-1, 1, 2000).astype(np.float32):
        assert abs(L.rzo_acos_f(float(x)) - math.acos(float(x))) <= 1.7e-4      # Mesa's polynomial
    assert L.rzo_acos_f(1.0) == 0.0 and L.rzo_rand_f(0.0, 0.0) == 0.0
    for x in (3.0e8, 2.5e9, 7.7e10):
        assert -1.0 <= L.rzo_sin_f(x) <= 1.0 and L.rzo_sin_f(x) == L.rzo_sin_f(x)


def test_hemisphere_direction_is_unit_and_on_the_normal_side():
    rng = np.random.default_rng(3)
    out = np.zeros(3, np.float32)
    for _ in range(300):
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        seed = rng.uniform(0, 5000, 2).astype(np.float32)
        rzo.lib().rzo_hemisphere_f(n.ctypes.data, seed.ctypes.data, out.ctypes.data)
        assert abs(np.linalg.norm(out.astype(np.float64)) - 1.0) < 1e-6
        assert float(np.dot(out.astype(np.float64), n.astype(np.float64))) > -1e-6


def test_hemisphere_at_seed_zero_is_the_normal():
    """at bounce 0 tempseed == (0,0): rand(0,0) = 0 -> theta = acos(1) = 0 -> dir = normal."""
    out = np.zeros(3, np.float32)
    n = np.array([0.0, 1.0, 0.0], np.float32)
    seed = np.zeros(2, np.float32)
    rzo.lib().rzo_hemisphere_f(n.ctypes.data, seed.ctypes.data, out.ctypes.data)
    # u = rand(0,0) = 0 -> theta = acos(sqrt(1)) = 0 -> local dir = (0,0,1) -> exactly the normal
    assert np.allclose(out, n, atol=1e-7)
