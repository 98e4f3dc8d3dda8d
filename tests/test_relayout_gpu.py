"""The device re-layout (rayzen_amd/csrc/hip/rz_relayout.hip) against the host re-layout it replaces (rz_context.hip:
build_view): the DevPair[] / DevTri[] bytes must be the same, so must every frame; inconsistent arrays must still be
rejected with RZ_ERR_BAD_SCENE (the device pass detects them, the host pass words the message)."""
import numpy as np
import pytest

from rayzen_amd import scene as S
from rayzen_amd.renderer import RayZenError, Renderer, frame_params
from helpers import oracle_render

pytestmark = pytest.mark.gpu
HOST_RELAYOUT = 4        # RZ_FLAG_HOST_RELAYOUT


def _layouts(sc, flags):
    r = Renderer(0, flags)
    r.upload_scene(sc)
    out = r.debug_read_layout(0), r.debug_read_layout(1)
    r.close()
    return out


@pytest.mark.parametrize("make", [
    lambda: S.cornell_scene(),
    lambda: S.bunny_scene(n=12, extras=True),
    lambda: S.instanced_scene(n=10, count=16, share_meshes=True),
    lambda: S.instanced_scene(n=6, count=4, share_meshes=False),
    lambda: S.bunny_scene(n=76),
    lambda: S.stress_scene(n=120),
])
def test_device_layout_is_byte_identical_to_the_host_layout(make):
    sc = make()
    dp, dt = _layouts(sc, 0)
    hp, ht = _layouts(sc, HOST_RELAYOUT)
    assert dp.size == hp.size and dp.size % 64 == 0 and (dp == hp).all()
    assert dt.size == ht.size and dt.size % 48 == 0 and (dt == ht).all()
    assert dp.size > 0 or sc.arrays[S.BIND_BLAS_NODES].shape[0] <= len(sc.arrays[S.BIND_INSTANCES])


def test_empty_meshes_and_leaf_roots():
    sc = S.Scene()
    e, q = sc.add_mesh(np.zeros(0, S.TRIANGLE)), sc.add_mesh(S.make_quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), 0))
    sc.add_object(e)
    sc.add_object(q, S.translate(S.identity(), (0, -1, -4)))      # 2 triangles: the root is a leaf
    sc.build()
    assert _layouts(sc, 0)[1].tobytes() == _layouts(sc, HOST_RELAYOUT)[1].tobytes()
    r = Renderer(0)
    r.upload_scene(sc)
    r.render_scene(sc, 40, 24, 2, 3)
    got = r.read_accum()
    r.close()
    assert (got.view(np.uint32) == oracle_render(sc, 40, 24, 2, 3).view(np.uint32)).all()


def test_inconsistent_arrays_are_still_rejected_and_the_context_recovers():
    sc = S.bunny_scene(n=8, extras=True)
    good = {b: sc.arrays[b].copy() for b in S.BINDING_DTYPES}
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, 32, 20, len(sc.lights), 3, 2))

    def expect_bad(binding, arr, word):
        r.upload(binding, arr)
        with pytest.raises(RayZenError) as e:
            r.render()
        assert e.value.code == -6 and word in str(e.value), str(e.value)
        r.upload(binding, good[binding])

    nodes = good[S.BIND_BLAS_NODES].copy()
    k = int(np.flatnonzero(nodes["count"] < 0)[3])
    nodes["leftFirst"][k] = len(nodes) + 5                              # child outside the array
    expect_bad(S.BIND_BLAS_NODES, nodes, "children")
    nodes = good[S.BIND_BLAS_NODES].copy()
    k = int(np.flatnonzero(nodes["count"] < 0)[5])
    nodes["leftFirst"][k] = 1                                           # a cycle: not a tree
    expect_bad(S.BIND_BLAS_NODES, nodes, "tree")
    nodes = good[S.BIND_BLAS_NODES].copy()
    k = int(np.flatnonzero(nodes["count"] > 0)[7])
    nodes["count"][k] = 99                                              # leaf too long
    expect_bad(S.BIND_BLAS_NODES, nodes, "leaf")
    idx = good[S.BIND_BLAS_INDICES].copy()
    idx[11] = 10 ** 7                                                   # triangle index outside the array
    expect_bad(S.BIND_BLAS_INDICES, idx, "triangle")
    tris = good[S.BIND_TRIANGLES].copy()
    tris["materialIndex"][40] = 77
    expect_bad(S.BIND_TRIANGLES, tris, "materialIndex")
    # fewer materials than the triangles name, geometry unchanged: caught by the device-side re-check
    r.render()
    r.upload(S.BIND_MATERIALS, good[S.BIND_MATERIALS][:2])
    with pytest.raises(RayZenError) as e:
        r.render()
    assert e.value.code == -6
    r.upload(S.BIND_MATERIALS, good[S.BIND_MATERIALS])
    r.render()
    got = r.read_accum()
    r.close()
    assert (got.view(np.uint32) == oracle_render(sc, 32, 20, 2, 3).view(np.uint32)).all()


def test_relayout_time_one_million_triangles():
    """Not a pass/fail timing: prints what the re-layout costs on each side (DESIGN.md quotes it)."""
    import time
    r = Renderer(0)
    sc = S.stress_scene(n=289, blas_builder=r)
    for flags, name in ((0, "device"), (HOST_RELAYOUT, "host")):
        q = Renderer(0, flags)
        times = []
        for _ in range(3):                           # the first pass pays for the allocations
            q.upload_scene(sc)
            q.set_frame(frame_params(sc.camera, 64, 36, len(sc.lights), 2, 1))
            t = time.perf_counter()
            q.render(); q.sync()                     # upload of the raw arrays + re-layout + one tiny frame
            times.append(time.perf_counter() - t)
        print(f"[relayout] {name}: {min(times) * 1e3:.1f} ms to the first frame after an upload of "
              f"{sc.arrays[S.BIND_TRIANGLES].shape[0]} triangles (best of 3; first {times[0] * 1e3:.1f})")
        q.close()
    r.close()
