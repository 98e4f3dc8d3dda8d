"""The pinned built-ins of oracle/rz_oracle_math.h: accuracy (they are meant to be sin/cos/acos, not just
deterministic), known answers, and the GLSL helper semantics."""
import math

import numpy as np

from oracle import rzo


def _ulp_err(got, want):
    got = np.float32(got)
    want32 = np.float32(want)
    if got == want32:
        return 0.0
    return abs(float(got) - want) / float(np.spacing(np.abs(want32)) or 1e-45)


def test_sin_cos_match_correctly_rounded_libm_over_the_rng_argument_range():
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


def test_acos_accuracy_and_endpoints():
    L = rzo.lib()
    rng = np.random.default_rng(5)
    for x in np.concatenate([rng.uniform(-1, 1, 4000), [0.0, 0.5, -0.5, 0.4999999, 0.99999994]]).astype(np.float32):
        assert _ulp_err(L.rzo_acos_f(float(x)), math.acos(float(x))) <= 1.0
    assert L.rzo_acos_f(1.0) == 0.0
    assert np.float32(L.rzo_acos_f(-1.0)) == np.float32(math.pi)
    assert math.isnan(L.rzo_acos_f(float("nan")))


def test_rand_is_fract_sin_dot():
    """FS:188-190 with the pinned sin: recompute in numpy float32, operation by operation."""
    L = rzo.lib()
    rng = np.random.default_rng(2)
    for _ in range(500):
        x, y = np.float32(rng.uniform(0, 3000)), np.float32(rng.uniform(0, 3000))
        d = np.float32(np.float32(x * np.float32(12.9898)) + np.float32(y * np.float32(78.233)))
        s = np.float32(L.rzo_sin_f(float(d)))
        p = np.float32(s * np.float32(43758.5453))
        want = np.float32(p - np.floor(p))
        got = np.float32(L.rzo_rand_f(float(x), float(y)))
        assert got == want and 0.0 <= got < 1.0


def test_rand_known_answers():
    """Values committed from this oracle (regression pin; the reference publishes none)."""
    L = rzo.lib()
    assert L.rzo_rand_f(0.3, 0.7) == 0.08203125
    assert L.rzo_rand_f(1234.5, 6789.1) == 0.755859375
    assert L.rzo_rand_f(0.0, 0.0) == 0.0


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
