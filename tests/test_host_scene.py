"""Host-side scene flatten (initializeSSBOs, main.cpp:941-1035), dynamic update (main.cpp:1138-1194) and the
OBJ reader (Mesh.cpp:6-50), against the oracle's literal restatements."""
import os

import numpy as np
import pytest

from oracle import rzo
from rayzen_amd import scene as S


def test_obj_loader_quirks(tmp_path):
    """'v '/'f ' lines only, tokens cut at the first '/', 1-based, fan triangulation, everything else ignored."""
    p = tmp_path / "m.obj"
    p.write_text("# comment\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nvn 0 0 1\nvt 0 0\n"
                 "s off\nusemtl x\nf 1 2 3\nf 1/5/2 2/1/1 3//7 4/2 5\nf 1 2\n")
    a = S.load_obj(str(p), 7)
    b = rzo.load_obj(str(p), 7)
    assert len(a) == 1 + 3 and a.tobytes() == b.tobytes()         # triangle + pentagon fan (3); 2-vertex face dropped
    assert (a["materialIndex"] == 7).all()
    assert a["v0"][1].tolist() == [0, 0, 0] and a["v1"][3].tolist() == [0, 1, 0] and a["v2"][3].tolist() == [0.5, 2, 0]


def test_obj_missing_file():
    import pytest
    with pytest.raises(FileNotFoundError):
        S.load_obj("/nonexistent/thing.obj", 0)


def _flatten_with_oracle(meshes, objects):
    """main.cpp:941-1035 restated with the oracle's builders: per object BLAS, offsets, world AABBs, TLAS."""
    tris, nodes, idx, inst, roots = [], [], [], [], []
    node_off = tri_off = tri_base = 0
    for i, (mid, m) in enumerate(objects):
        n, ix = rzo.build_blas(meshes[mid])
        tris.append(meshes[mid]); nodes.append(n); idx.append(ix)
        mn, mx = rzo.world_bounds(n[0], m)
        r = np.zeros(1, rzo.NODE)
        r[0] = n[0]
        r["bmin"], r["bmax"] = mn, mx
        roots.append(r)
        rec = np.zeros(1, rzo.INSTANCE)
        rec["blasNodeOffset"], rec["blasTriOffset"], rec["meshIndex"], rec["globalTriOffset"] = node_off, tri_off, i, tri_base
        rec["transform"] = m
        inst.append(rec)
        node_off += len(n); tri_off += len(ix); tri_base += len(meshes[mid])
    tn, ti = rzo.build_tlas(np.concatenate(roots))
    return np.concatenate(tris), np.concatenate(nodes), np.concatenate(idx), np.concatenate(inst), tn, ti


def test_flatten_matches_literal_restatement():
    meshes = [S.make_cube(0), S.make_blob(5, 2.0, 1), S.make_quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), 4)]
    objs = [(0, S.translate(S.scale(S.identity(), (8, .5, 8)), (0, -3, 0))), (1, S.translate(S.identity(), (-4, 0, 0))),
            (1, S.translate(S.identity(), (4, 0, 0))), (2, S.identity()), (0, S.rotate(S.identity(), 0.4, (0, 1, 0)))]
    sc = S.Scene()
    ids = [sc.add_mesh(m) for m in meshes]
    for mid, m in objs:
        sc.add_object(ids[mid], m)
    sc.build()
    tris, nodes, idx, inst, tn, ti = _flatten_with_oracle(meshes, objs)
    a = sc.arrays
    assert a[S.BIND_TRIANGLES].tobytes() == tris.tobytes()
    assert a[S.BIND_BLAS_NODES].tobytes() == nodes.tobytes()
    assert a[S.BIND_BLAS_INDICES].tobytes() == idx.tobytes()
    assert a[S.BIND_TLAS_NODES].tobytes() == tn.tobytes() and a[S.BIND_TLAS_INDICES].tobytes() == ti.tobytes()
    got = a[S.BIND_INSTANCES]
    for f in ("blasNodeOffset", "blasTriOffset", "meshIndex", "globalTriOffset", "transform"):
        assert (got[f] == inst[f]).all(), f
    for i, (_, m) in enumerate(objs):      # inverseTransform really is the inverse
        M, Mi = m.reshape(4, 4).T.astype(np.float64), got["inverseTransform"][i].reshape(4, 4).T.astype(np.float64)
        assert np.allclose(M @ Mi, np.eye(4), atol=1e-5)


def test_share_meshes_points_instances_at_one_copy():
    sc = S.Scene()
    cube, blob = sc.add_mesh(S.make_cube(0)), sc.add_mesh(S.make_blob(4, 2.0, 1))
    for k in range(3):
        sc.add_object(blob, S.translate(S.identity(), (3.0 * k, 0, 0)))
    sc.add_object(cube)
    sc.build(share_meshes=True)
    inst = sc.arrays[S.BIND_INSTANCES]
    assert len(set(inst["blasNodeOffset"][:3])) == 1 and len(set(inst["globalTriOffset"][:3])) == 1
    assert len(sc.arrays[S.BIND_TRIANGLES]) == 12 * 16 + 12
    dup = S.Scene()
    c2, b2 = dup.add_mesh(S.make_cube(0)), dup.add_mesh(S.make_blob(4, 2.0, 1))
    for k in range(3):
        dup.add_object(b2, S.translate(S.identity(), (3.0 * k, 0, 0)))
    dup.add_object(c2)
    dup.build(share_meshes=False)
    assert len(dup.arrays[S.BIND_TRIANGLES]) == 3 * 12 * 16 + 12        # the reference duplicates per object


def test_update_dynamic_rebuilds_only_instances_and_tlas():
    sc = S.instanced_scene(n=4, count=4)
    before = {b: sc.arrays[b].copy() for b in sc.arrays}
    for oid, t in zip(sc.instance_ids, S.instanced_transforms(7, 4)):
        sc.set_transform(oid, t)
    sc.update_dynamic()
    for b in (S.BIND_TRIANGLES, S.BIND_BLAS_NODES, S.BIND_BLAS_INDICES):
        assert sc.arrays[b].tobytes() == before[b].tobytes()
    assert sc.arrays[S.BIND_INSTANCES].tobytes() != before[S.BIND_INSTANCES].tobytes()
    assert len(sc.arrays[S.BIND_TLAS_NODES]) == len(before[S.BIND_TLAS_NODES])      # 2*I-1, fixed
    # equals a from-scratch build with the new transforms
    fresh = S.Scene()
    floor, bunny = fresh.add_mesh(S.make_cube(4)), fresh.add_mesh(S.make_blob(4, 2.8, 0))
    fresh.add_object(floor, S.translate(S.scale(S.identity(), (8.0, 0.5, 8.0)), (0.0, -3.0, 0.0)))
    for t in S.instanced_transforms(7, 4):
        fresh.add_object(bunny, t)
    fresh.build(share_meshes=True)
    for b in (S.BIND_INSTANCES, S.BIND_TLAS_NODES, S.BIND_TLAS_INDICES):
        assert sc.arrays[b].tobytes() == fresh.arrays[b].tobytes()


def test_camera_matrices_are_consistent():
    cam = S.Camera(position=(1, 2, 3), target=(0.1, -0.2, -1), fov=70.0, aspect=16 / 9)
    V, P = cam.view.reshape(4, 4).T.astype(np.float64), cam.proj.reshape(4, 4).T.astype(np.float64)
    assert np.allclose(V @ cam.inv_view.reshape(4, 4).T, np.eye(4), atol=1e-5)
    assert np.allclose(P @ cam.inv_proj.reshape(4, 4).T, np.eye(4), atol=1e-4)
    assert np.allclose((V @ np.array([1, 2, 3, 1.0]))[:3], 0, atol=1e-5)             # eye maps to the origin
    f = 1.0 / np.tan(np.radians(70.0) / 2)
    assert abs(P[1, 1] - f) < 1e-5 and abs(P[0, 0] - f / (16 / 9)) < 1e-5 and P[3, 2] == -1.0


def test_scene_cache_round_trip_in_rayzen_format(tmp_path):
    """main.cpp:94-115: every cache file is a native size_t count followed by the raw POD array."""
    import struct
    sc = S.bunny_scene(n=5, extras=True)
    d = str(tmp_path / "bvh_cache" / "v2")
    sc.save_cache(d)
    names = {"triangles": S.BIND_TRIANGLES, "blasnodes": S.BIND_BLAS_NODES, "blastris": S.BIND_BLAS_INDICES,
             "instances": S.BIND_INSTANCES, "tlasnodes": S.BIND_TLAS_NODES, "tlastris": S.BIND_TLAS_INDICES}
    for name, b in names.items():
        raw = open(os.path.join(d, f"ssbo_v2_{name}.bin"), "rb").read()
        (n,) = struct.unpack("<Q", raw[:8])
        assert n == len(sc.arrays[b]) and raw[8:] == sc.arrays[b].tobytes()
    other = S.Scene().load_cache(d)
    for b in names.values():
        assert other.arrays[b].tobytes() == sc.arrays[b].tobytes()
    assert (other.max_blas_depth, other.tlas_depth) == (sc.max_blas_depth, sc.tlas_depth)
    # a truncated file is refused, not trusted
    p = os.path.join(d, "ssbo_v2_blasnodes.bin")
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:len(raw) // 2])
    import pytest
    with pytest.raises(OSError):
        S.Scene().load_cache(d)


def test_blas_builder_hook_is_used_and_its_failure_is_reported():
    """rzh_scene_set_blas_builder: scene assembly calls a function with rz_build_blas's signature for every distinct
    mesh (here the host builder wrapped as a callback, so no GPU is needed) and reports its failure instead of
    silently building on the host."""
    import ctypes as C
    from rayzen_amd import _lib
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                        C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_float))
    calls = []

    def builder(ctx, tris, n, nodes_out, cap, idx_out, n_nodes, depth, ms):
        calls.append(int(n))
        if fail[0]:
            return -3
        d = C.c_int(0)
        nn = _lib.host().rzh_build_blas(tris, n, nodes_out, idx_out, C.byref(d))
        assert 0 < nn <= cap
        n_nodes[0] = nn
        return 0

    fail = [False]
    cb = proto(builder)
    ref = S.bunny_scene(n=6, extras=True)
    s = S.bunny_scene(n=6, extras=True)
    assert _lib.host().rzh_scene_set_blas_builder(s._h, C.cast(cb, C.c_void_p), None) == 0
    s.build()
    assert sorted(calls) == sorted(int(m) for m in (12, 12 * 36, 12 * 16, 12))     # floor, bunny, glass blob, mirror cube
    for b in S.GEOMETRY_BINDINGS:
        assert s.arrays[b].tobytes() == ref.arrays[b].tobytes()
    fail[0] = True
    with pytest.raises(RuntimeError, match="device BLAS builder failed"):
        s.build()
    s.set_blas_builder(None)
    s.build()
    for b in S.GEOMETRY_BINDINGS:
        assert s.arrays[b].tobytes() == ref.arrays[b].tobytes()


def _dynamic_scene(transforms):
    """floor + one blob instance per transform (each object its own mesh copy: the reference never shares)."""
    sc = S.Scene()
    floor, blob = sc.add_mesh(S.make_cube(4)), sc.add_mesh(S.make_blob(4, 2.8, 0))
    sc.add_object(floor, S.translate(S.scale(S.identity(), (8.0, 0.5, 8.0)), (0.0, -3.0, 0.0)))
    for t in transforms:
        sc.add_object(blob, t)
    return sc


def _files(d):
    return sorted(os.listdir(d))


def test_full_disk_cache_flow_of_initializeSSBOs(tmp_path):
    """RayZen/src/main.cpp:897-1060, step for step: the ssbo_v2_* set, per-object mesh<i>.nodes/.tris, scene_tlas.* +
    instances.bin, all `size_t count + raw POD`; what is on disk decides what is rebuilt."""
    import struct
    d = str(tmp_path / "bvh_cache" / "v2")
    xf = S.instanced_transforms(0, 3)
    ref = _dynamic_scene(xf).build(share_meshes=False)              # no cache: the expected arrays
    first = _dynamic_scene(xf).build_cached(d)
    assert first.cache_report == dict(ssbo_loaded=False, ssbo_invalidated=False, blas_loaded=0, blas_built=4, tlas_loaded=False)
    for b in S.GEOMETRY_BINDINGS:
        assert first.arrays[b].tobytes() == ref.arrays[b].tobytes(), b
    want = ["instances.bin", "scene_tlas.nodes.bin", "scene_tlas.tris.bin"]
    want += [f"mesh{i}.{k}.bin" for i in range(4) for k in ("nodes", "tris")]
    want += [f"ssbo_v2_{n}.bin" for n in ("triangles", "blasnodes", "blastris", "instances", "tlasnodes", "tlastris")]
    assert _files(d) == sorted(want)
    # byte layout of the per-mesh / TLAS / instance files (main.cpp:94-133)
    inst = ref.arrays[S.BIND_INSTANCES]
    bn, bi = ref.arrays[S.BIND_BLAS_NODES], ref.arrays[S.BIND_BLAS_INDICES]
    for i in range(4):
        n0 = int(inst["blasNodeOffset"][i]); n1 = int(inst["blasNodeOffset"][i + 1]) if i < 3 else len(bn)
        t0 = int(inst["blasTriOffset"][i]); t1 = int(inst["blasTriOffset"][i + 1]) if i < 3 else len(bi)
        raw = open(os.path.join(d, f"mesh{i}.nodes.bin"), "rb").read()
        assert struct.unpack("<Q", raw[:8])[0] == n1 - n0 and raw[8:] == bn[n0:n1].tobytes()
        raw = open(os.path.join(d, f"mesh{i}.tris.bin"), "rb").read()
        assert struct.unpack("<Q", raw[:8])[0] == t1 - t0 and raw[8:] == bi[t0:t1].tobytes()
    raw = open(os.path.join(d, "scene_tlas.nodes.bin"), "rb").read()
    assert raw[8:] == ref.arrays[S.BIND_TLAS_NODES].tobytes()
    raw = open(os.path.join(d, "scene_tlas.tris.bin"), "rb").read()
    assert raw[8:] == ref.arrays[S.BIND_TLAS_INDICES].tobytes()
    raw = open(os.path.join(d, "instances.bin"), "rb").read()
    assert struct.unpack("<Q", raw[:8])[0] == 4 and raw[8:] == inst.tobytes() and len(raw) == 8 + 4 * 144

    # second start: the ssbo set is used, nothing is built
    again = _dynamic_scene(xf).build_cached(d)
    assert again.cache_report["ssbo_loaded"] and again.cache_report["blas_built"] == 0
    for b in S.GEOMETRY_BINDINGS:
        assert again.arrays[b].tobytes() == ref.arrays[b].tobytes(), b
    assert (again.max_blas_depth, again.tlas_depth) == (ref.max_blas_depth, ref.tlas_depth)

    # ssbo set gone: every BLAS + the TLAS + the instances come from their own files, and the ssbo set is rewritten
    for f in _files(d):
        if f.startswith("ssbo_v2_"):
            os.remove(os.path.join(d, f))
    third = _dynamic_scene(xf).build_cached(d)
    assert third.cache_report == dict(ssbo_loaded=False, ssbo_invalidated=False, blas_loaded=4, blas_built=0, tlas_loaded=True)
    for b in S.GEOMETRY_BINDINGS:
        assert third.arrays[b].tobytes() == ref.arrays[b].tobytes(), b
    assert "ssbo_v2_triangles.bin" in _files(d)

    # --rebuild-bvh ignores everything on disk
    forced = _dynamic_scene(xf).build_cached(d, force_rebuild=True)
    assert forced.cache_report == dict(ssbo_loaded=False, ssbo_invalidated=False, blas_loaded=0, blas_built=4, tlas_loaded=False)


def test_disk_cache_invalidation_and_the_reference_s_stale_data_quirks(tmp_path):
    d = str(tmp_path / "c")
    xf = S.instanced_transforms(0, 3)
    _dynamic_scene(xf).build_cached(d)
    # (1) a changed object COUNT is the only thing that invalidates the ssbo set (main.cpp:929-934); the BLAS of the
    #     objects that still exist come from their mesh<i> files, the new object's is built, the TLAS is rebuilt
    xf4 = S.instanced_transforms(0, 4)
    grown = _dynamic_scene(xf4).build_cached(d)
    assert grown.cache_report == dict(ssbo_loaded=False, ssbo_invalidated=True, blas_loaded=4, blas_built=1, tlas_loaded=False)
    fresh = _dynamic_scene(xf4).build(share_meshes=False)
    for b in S.GEOMETRY_BINDINGS:
        assert grown.arrays[b].tobytes() == fresh.arrays[b].tobytes(), b
    # (2) same count, moved objects, ssbo set present: transforms / inverses are refreshed from the scene
    #     (main.cpp:1054-1060) but the cached TLAS is used as it is -- stale, exactly like the reference
    moved = S.instanced_transforms(9, 4)
    stale = _dynamic_scene(moved).build_cached(d)
    assert stale.cache_report["ssbo_loaded"]
    want = _dynamic_scene(moved).build(share_meshes=False)
    assert stale.arrays[S.BIND_INSTANCES].tobytes() == want.arrays[S.BIND_INSTANCES].tobytes()
    assert stale.arrays[S.BIND_TLAS_NODES].tobytes() == grown.arrays[S.BIND_TLAS_NODES].tobytes()      # the OLD boxes
    assert stale.arrays[S.BIND_TLAS_NODES].tobytes() != want.arrays[S.BIND_TLAS_NODES].tobytes()
    # (3) ssbo set gone, all BLAS cached: scene_tlas + instances.bin are loaded, and instances.bin REPLACES the records
    #     just assembled, transforms included (main.cpp:1013: loadBVHInstancesFromFile into meshInstances)
    for f in _files(d):
        if f.startswith("ssbo_v2_"):
            os.remove(os.path.join(d, f))
    old = _dynamic_scene(moved).build_cached(d)
    assert old.cache_report["tlas_loaded"] and old.cache_report["blas_loaded"] == 5
    assert old.arrays[S.BIND_INSTANCES].tobytes() == grown.arrays[S.BIND_INSTANCES].tobytes()          # the OLD transforms
    # (4) a truncated per-mesh file is not trusted: that BLAS is rebuilt, and then the TLAS is too
    p = os.path.join(d, "mesh2.nodes.bin")
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:len(raw) // 2])
    for f in _files(d):
        if f.startswith("ssbo_v2_"):
            os.remove(os.path.join(d, f))
    healed = _dynamic_scene(moved).build_cached(d)
    assert healed.cache_report == dict(ssbo_loaded=False, ssbo_invalidated=False, blas_loaded=4, blas_built=1, tlas_loaded=False)
    for b in S.GEOMETRY_BINDINGS:
        assert healed.arrays[b].tobytes() == want.arrays[b].tobytes(), b


def test_obj_mesh_replaces_the_stand_in_and_is_fitted_in_object_space(reference_dir):
    """a real OBJ dropped into assets/ is rendered instead of the procedural mesh.  Its vertices (not its
    transform) are scaled: the shader's |a| < 1e-4 cull works in object space (FS:396)."""
    path = os.path.join(reference_dir, "meshes", "monkey.obj")
    sc = S.bunny_scene(obj_path=path, radius=2.8)
    tris = sc.arrays[S.BIND_TRIANGLES]
    assert len(tris) == 12 + 968                                        # floor cube + Suzanne
    v = np.stack([tris["v0"][12:], tris["v1"][12:], tris["v2"][12:]]).reshape(-1, 3)
    half = (v.max(axis=0) - v.min(axis=0)) * 0.5
    assert abs(float(half.max()) - 2.8) < 1e-4 and np.allclose((v.max(axis=0) + v.min(axis=0)) * 0.5, 0, atol=1e-5)
    inst = sc.arrays[S.BIND_INSTANCES]
    assert np.allclose(inst["transform"][1].reshape(4, 4).T[:3, :3], np.eye(3))   # translation only: no scale in the transform


def test_named_configs_keep_the_camera_outside_every_object():
    """Every named workload's camera stands OUTSIDE every instance's world box (RayZen's does: camera (0, 0, 3), monkey D
    centred at (0, 0, 4) reaches z = 3.148, RayZen/src/main.cpp:331-339, 383 + meshes/monkey.obj's extents).  Round 4's `ref`
    rows were rendered from INSIDE a too-large stand-in mesh and described a closed room."""
    import numpy as np
    from rayzen_amd import scene as S
    for name in ("c1", "c2", "c2close", "c2g", "c4", "ref", "ref64"):
        sc = S.named_config(name)[0]
        assert S.camera_clearance(sc) > 0.1, (name, S.camera_clearance(sc))
    # the stand-in has Suzanne's box, exactly, and RayZen's triangle budget
    sc = S.reference_scene()
    inst, nodes = sc.arrays[S.BIND_INSTANCES], sc.arrays[S.BIND_BLAS_NODES]
    root = nodes[int(inst[1]["blasNodeOffset"])]
    assert np.allclose(root["boundsMax"], S.SUZANNE_HALF_EXTENTS, atol=1e-6) and np.allclose(root["boundsMin"], [-x for x in S.SUZANNE_HALF_EXTENTS], atol=1e-6)
    assert sc.arrays[S.BIND_TRIANGLES].shape[0] == 12 + 5 * 972
