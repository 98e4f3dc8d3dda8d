"""Host BVH builders (product, O(N log N)) against the oracle's literal restatement of RayZen/src/BVH.cpp
(O(N log^2 N)): byte-identical node and index arrays; plus the node counts the survey measured by running
the reference's own BVH.cpp and structural invariants."""
import os

import numpy as np
import pytest

from oracle import rzo
from rayzen_amd import scene as S


def _soup(n, seed, scale=1.0, dup=False):
    rng = np.random.default_rng(seed)
    t = np.zeros(n, S.TRIANGLE)
    c = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    if dup:
        c[n // 2:] = c[:n - n // 2]           # duplicate centroids: the (key, id) tie-break must decide
    for k in ("v0", "v1", "v2"):
        t[k] = c + rng.uniform(-scale, scale, (n, 3)).astype(np.float32)
    if dup:
        t["v0"][n // 2:], t["v1"][n // 2:], t["v2"][n // 2:] = (t["v0"][:n - n // 2], t["v1"][:n - n // 2],
                                                                t["v2"][:n - n // 2])
    t["materialIndex"] = rng.integers(0, 5, n)
    return t


def _check_tree(nodes, idx, tris):
    n = len(tris)
    assert sorted(idx.tolist()) == list(range(n))            # a permutation
    seen = np.zeros(n, bool)
    stack = [0]
    visited = 0
    while stack:
        i = stack.pop()
        visited += 1
        nd = nodes[i]
        if nd["count"] >= 0:
            assert 0 <= nd["count"] <= 4
            ids = idx[nd["leftFirst"]:nd["leftFirst"] + nd["count"]]
            assert not seen[ids].any()
            seen[ids] = True
            if len(ids):
                v = np.concatenate([tris[k][ids] for k in ("v0", "v1", "v2")])
                assert (v.min(axis=0) == nd["boundsMin"]).all() and (v.max(axis=0) == nd["boundsMax"]).all()
        else:
            L = nd["leftFirst"]
            assert L % 2 == 1                                 # children allocated as adjacent pairs after the root
            for ch in (L, L + 1):
                assert (nodes[ch]["boundsMin"] >= nd["boundsMin"]).all() and (nodes[ch]["boundsMax"] <= nd["boundsMax"]).all()
            stack += [L, L + 1]
    assert visited == len(nodes) and seen.all()


@pytest.mark.parametrize("make", [
    lambda: S.make_cube(0),
    lambda: S.make_blob(6, 2.8, 0),
    lambda: S.make_blob(20, 2.8, 1),
    lambda: _soup(1, 0), lambda: _soup(4, 1), lambda: _soup(5, 2), lambda: _soup(777, 3),
    lambda: _soup(3000, 4, scale=0.05), lambda: _soup(600, 5, dup=True),
    lambda: np.repeat(_soup(1, 6), 37),                      # 37 identical triangles: SAH cost ties everywhere
])
def test_host_builder_is_byte_identical_to_the_literal_restatement(make):
    tris = make()
    n1, i1, depth = S.build_blas(tris)
    n2, i2 = rzo.build_blas(tris)
    assert n1.tobytes() == n2.tobytes()
    assert i1.tobytes() == i2.tobytes()
    _check_tree(n1, i1, tris)
    assert depth >= 1


def test_cube_matches_the_reference_run_recorded_by_the_survey():
    """reference buildBLAS on cube.obj -> 9 nodes."""
    nodes, idx, depth = S.build_blas(S.make_cube(0))
    assert len(nodes) == 9 and (nodes["count"] > 0).sum() == 5 and depth == 5   # "depth 4" counted in edges


def test_monkey_matches_the_reference_run_recorded_by_the_survey(reference_dir):
    """monkey.obj (968 tris) -> 625 nodes, 313 leaves, depth 11, 3.09 tris/leaf."""
    path = os.path.join(reference_dir, "meshes", "monkey.obj")
    tris = S.load_obj(path, 1)
    assert len(tris) == 968
    assert tris.tobytes() == rzo.load_obj(path, 1).tobytes()
    nodes, idx, depth = S.build_blas(tris)
    leaves = nodes["count"][nodes["count"] > 0]
    assert (len(nodes), len(leaves), depth - 1) == (625, 313, 11)
    assert abs(leaves.mean() - 3.09) < 0.01
    n2, i2 = rzo.build_blas(tris)
    assert nodes.tobytes() == n2.tobytes() and idx.tobytes() == i2.tobytes()


def test_synthetic_cube_is_the_reference_asset(reference_dir):
    assert S.make_cube(3).tobytes() == S.load_obj(os.path.join(reference_dir, "meshes", "cube.obj"), 3).tobytes()


def test_empty_mesh_gives_an_empty_leaf_root_with_inverted_bounds():
    """BVH.cpp:12-13,115-118: count 0, bounds (+FLT_MAX, -FLT_MAX) -> no ray can enter."""
    nodes, idx, depth = S.build_blas(np.zeros(0, S.TRIANGLE))
    assert len(nodes) == 1 and nodes[0]["count"] == 0
    assert (nodes[0]["boundsMin"] > 1e38).all() and (nodes[0]["boundsMax"] < -1e38).all()
    n2, _ = rzo.build_blas(np.zeros(0, S.TRIANGLE))
    assert nodes.tobytes() == n2.tobytes()


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (3, 2), (16, 3), (50, 4)])
def test_tlas_builder_matches(n, seed):
    rng = np.random.default_rng(seed)
    roots = np.zeros(n, S.BVH_NODE)
    lo = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    roots["boundsMin"], roots["boundsMax"] = lo, lo + rng.uniform(0.1, 4, (n, 3)).astype(np.float32)
    a_nodes, a_idx = S.build_tlas(roots)
    b_nodes, b_idx = rzo.build_tlas(roots)
    assert a_nodes.tobytes() == b_nodes.tobytes() and a_idx.tobytes() == b_idx.tobytes()
    assert len(a_nodes) == 2 * n - 1 and sorted(a_idx.tolist()) == list(range(n))     # 1 instance per leaf
    assert ((a_nodes["count"] == 1) | (a_nodes["count"] == -1)).all()


def test_tlas_identical_boxes_uses_the_count_over_two_fallback():
    roots = np.zeros(6, S.BVH_NODE)
    roots["boundsMin"], roots["boundsMax"] = (-1, -1, -1), (1, 1, 1)
    a_nodes, a_idx = S.build_tlas(roots)
    b_nodes, b_idx = rzo.build_tlas(roots)
    assert a_nodes.tobytes() == b_nodes.tobytes() and a_idx.tolist() == b_idx.tolist()


def test_world_bounds():
    from rayzen_amd import _lib
    root = np.zeros(1, S.BVH_NODE)
    root["boundsMin"], root["boundsMax"] = (-1, -2, -3), (1, 2, 3)
    m = S.rotate(S.translate(S.scale(S.identity(), (2, 0.5, 1)), (1, 2, 3)), 0.7, (0.3, 1.0, 0.2))
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    _lib.host().rzh_world_bounds(root.ctypes.data, m.ctypes.data, mn.ctypes.data, mx.ctypes.data)
    omn, omx = rzo.world_bounds(root[0], m)
    assert (mn == omn).all() and (mx == omx).all()
    M = m.reshape(4, 4).T.astype(np.float64)
    corners = np.array([[x, y, z, 1.0] for x in (-1, 1) for y in (-2, 2) for z in (-3, 3)])
    w = (M @ corners.T).T[:, :3]
    assert np.allclose(mn, w.min(axis=0), atol=1e-5) and np.allclose(mx, w.max(axis=0), atol=1e-5)


@pytest.mark.slow
def test_bunny_sized_mesh_byte_identical_and_fast():
    tris = S.make_blob(76, 2.8, 0)
    assert len(tris) == 69312
    n1, i1, depth = S.build_blas(tris)
    n2, i2 = rzo.build_blas(tris)
    assert n1.tobytes() == n2.tobytes() and i1.tobytes() == i2.tobytes()
    assert depth <= 64              # the shader's stack[64] (FS:422) would hold it
