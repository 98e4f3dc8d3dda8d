"""rz_build_blas (device) vs BVH::buildBLAS as restated by the oracle (oracle/rz_oracle_bvh.c, literal O(N log^2 N)
sort-per-node version of RayZen/src/BVH.cpp:22-175) and by the host library: nodes and indices byte for byte."""
import numpy as np
import pytest

from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer
from oracle import rzo

pytestmark = pytest.mark.gpu


def soup(n, seed, spread=4.0, size=0.3):
    rng = np.random.default_rng(seed)
    t = np.zeros(n, S.TRIANGLE)
    c = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
    for k in ("v0", "v1", "v2"):
        t[k] = c + rng.uniform(-size, size, (n, 3)).astype(np.float32)
    t["materialIndex"] = 0
    return t


def check(r, tris, oracle=True):
    nodes, idx, depth, ms = r.build_blas(tris)
    hn, hi, hd = S.build_blas(tris)
    assert nodes.shape == hn.shape, (nodes.shape, hn.shape)
    assert nodes.tobytes() == hn.tobytes()
    assert idx.tobytes() == hi.tobytes()
    assert depth == hd
    if oracle:
        on, oi = rzo.build_blas(tris)
        assert nodes.tobytes() == on.tobytes()
        assert idx.tobytes() == oi.tobytes()
    return nodes, idx, ms


@pytest.fixture(scope="module")
def r():
    rr = Renderer(0)
    yield rr
    rr.close()


@pytest.mark.parametrize("n", [1, 2, 4, 5, 7, 33, 257, 2049, 5000])
def test_random_soups(r, n):
    check(r, soup(n, n))


def test_cube_blob_and_bunny_stand_in(r):
    check(r, S.make_cube(0))
    check(r, S.make_blob(12, 1.0, 0))
    check(r, S.make_blob(40, 1.0, 0, seed=3))
    nodes, idx, ms = check(r, S.make_blob(76, 1.0, 0), oracle=False)      # C2's mesh (69 312 triangles)
    assert sorted(idx.tolist()) == list(range(idx.shape[0]))


def test_duplicate_centroids_fall_back_to_triangle_id_order(r):
    t = soup(600, 5)
    t[200:400] = t[0:200]                      # exact duplicates: ties on every axis
    check(r, t)
    g = soup(64, 6)
    for k in ("v0", "v1", "v2"):               # centroids on a coarse grid: many equal keys per axis
        g[k] = np.round(g[k] * 2) / 2
    check(r, np.concatenate([g] * 5))


def test_signed_zeros_keep_the_sign_computeBounds_keeps(r):
    t = soup(300, 7, spread=1.0, size=0.5)
    rng = np.random.default_rng(8)
    for k in ("v0", "v1", "v2"):
        v = t[k]
        m = rng.random(v.shape) < 0.3
        v[m] = np.where(rng.random(m.sum()) < 0.5, np.float32(0.0), np.float32(-0.0))
        v[:, 1] = np.where(v[:, 1] < 0, np.float32(-0.0), v[:, 1])   # a floor at y = -0 / +0
        t[k] = v
    check(r, t)


def test_degenerate_meshes_take_the_midpoint_fallback(r):
    # all triangles identical and of zero area: parentArea == 0, every cost is 0/1e-6 = 0 -> SAH still splits at i=1;
    # a huge box makes the costs overflow to inf so that no split is accepted and BVH.cpp:135-149 runs
    t = np.zeros(37, S.TRIANGLE)
    check(r, t)
    big = soup(50, 9, spread=1e19, size=1e18)
    check(r, big)
    mixed = np.concatenate([soup(200, 10), big])
    check(r, mixed)


def test_empty_mesh(r):
    nodes, idx, depth, ms = r.build_blas(np.zeros(0, S.TRIANGLE))
    hn, hi, hd = S.build_blas(np.zeros(0, S.TRIANGLE))
    assert nodes.tobytes() == hn.tobytes() and idx.shape == (0,)


def test_a_larger_mesh_and_the_scene_built_from_it_renders_identically(r):
    t = S.make_blob(150, 1.0, 0)               # 270 000 triangles
    nodes, idx, ms = check(r, t, oracle=False)
    assert ms > 0


def test_scene_build_with_the_device_builder_gives_the_same_buffers(r):
    host = S.bunny_scene(n=20, extras=True)
    dev = S.bunny_scene(n=20, extras=True, blas_builder=r)
    for b in S.GEOMETRY_BINDINGS:
        assert dev.arrays[b].tobytes() == host.arrays[b].tobytes(), b
    assert (dev.max_blas_depth, dev.tlas_depth) == (host.max_blas_depth, host.tlas_depth)
    dev.set_blas_builder(None)              # back to the host builder: still the same
    dev.build()
    for b in S.GEOMETRY_BINDINGS:
        assert dev.arrays[b].tobytes() == host.arrays[b].tobytes(), b


def test_rz_build_blas_argument_errors(r):
    import ctypes as C
    from rayzen_amd.renderer import RayZenError
    L, c = r._L, r._c
    t = soup(100, 1)
    nodes = np.zeros(199, S.BVH_NODE)
    idx = np.zeros(100, np.int32)
    nn = C.c_size_t(0)
    assert L.rz_build_blas(c, t.ctypes.data, 100, nodes.ctypes.data, 50, idx.ctypes.data, C.byref(nn), None, None) == -7   # RZ_ERR_BUFFER_SIZE
    assert L.rz_build_blas(c, None, 100, nodes.ctypes.data, 199, idx.ctypes.data, C.byref(nn), None, None) == -1
    assert L.rz_build_blas(c, t.ctypes.data, 100, None, 199, idx.ctypes.data, C.byref(nn), None, None) == -1
    assert L.rz_build_blas(None, t.ctypes.data, 100, nodes.ctypes.data, 199, idx.ctypes.data, C.byref(nn), None, None) == -1
    assert L.rz_build_blas(c, t.ctypes.data, 100, nodes.ctypes.data, 199, idx.ctypes.data, C.byref(nn), None, None) == 0
    hn, hi, _ = S.build_blas(t)
    assert nodes[:nn.value].tobytes() == hn.tobytes() and idx.tobytes() == hi.tobytes()
