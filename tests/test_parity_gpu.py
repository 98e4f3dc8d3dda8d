"""GPU parity proper: the HIP path (through the C-ABI) against the CPU oracle on the same scene arrays.

Bar: accumulation-buffer Linf < 1e-4 per channel (BASELINE.json north_star).  By construction the
arithmetic is identical, so the expectation is Linf == 0 and every pixel bit-identical."""
import numpy as np
import pytest

from rayzen_amd import scene as S
from helpers import hip_render, oracle_render, linf, mismatch_report

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star: image L-inf error < 1e-4 on the accumulation buffer


BACKENDS = ["auto", "pixel"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_c1_cornell_256_4spp_1bounce(backend):
    sc = S.cornell_scene()
    ref, rc = oracle_render(sc, 256, 256, 4, 1, want_counters=True)
    gpu, gc = hip_render(sc, 256, 256, 4, 1, counted=True, backend=backend)
    assert linf(gpu, ref) < TOL, mismatch_report(gpu, ref)
    assert gc == rc, (gc, rc)          # the counted kernel tallies exactly the oracle's memory touches
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(gpu, ref)


@pytest.mark.parametrize("backend", BACKENDS)
def test_c1_uncounted_equals_counted(backend):
    sc = S.cornell_scene()
    a = hip_render(sc, 256, 256, 4, 1, backend=backend)
    b, _ = hip_render(sc, 256, 256, 4, 1, counted=True, backend=backend)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("bounces,spp", [(4, 2), (5, 3), (8, 1)])
def test_bunny_small_frame(bounces, spp, backend):
    sc = S.bunny_scene(n=24, extras=True)
    W, H = 160, 90
    ref, rc = oracle_render(sc, W, H, spp, bounces, want_counters=True)
    gpu, gc = hip_render(sc, W, H, spp, bounces, counted=True, backend=backend)
    assert linf(gpu, ref) < TOL, mismatch_report(gpu, ref)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(gpu, ref)
    assert gc == rc
