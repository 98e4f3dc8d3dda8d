"""rz_build_geometry: BLAS built on the device and LEFT there as the context's bindings (no host round trip), against
the host assembly (SceneBuffers::build with shared meshes): same node / index / triangle arrays, same frames."""
import numpy as np
import pytest

from rayzen_amd import scene as S
from rayzen_amd.renderer import RayZenError, Renderer, frame_params
from helpers import oracle_render

pytestmark = pytest.mark.gpu


def _scene_parts(n, count):
    meshes = [S.make_cube(4), S.make_blob(n, 2.8, 0), np.zeros(0, S.TRIANGLE), S.make_blob(max(4, n // 3), 1.2, 3, seed=7)]
    objects = [(0, S.translate(S.scale(S.identity(), (8.0, 0.5, 8.0)), (0.0, -3.0, 0.0)))]
    objects += [(1, t) for t in S.instanced_transforms(2, count)]
    objects += [(2, S.identity()), (3, S.translate(S.identity(), (4.5, 0.6, 3.0)))]
    return meshes, objects


def _host_scene(meshes, objects, camera):
    sc = S.Scene(camera=camera)
    ids = [sc.add_mesh(m) for m in meshes]
    for mi, xf in objects:
        sc.add_object(ids[mi], xf)
    return sc.build(share_meshes=True)


@pytest.mark.parametrize("n,count", [(6, 4), (24, 9)])
def test_device_built_geometry_equals_the_host_assembly(n, count):
    cam = S.Camera(position=(0.0, 5.0, 12.0), target=(0.0, -0.35, -1.0), aspect=16 / 9)
    meshes, objects = _scene_parts(n, count)
    ref = _host_scene(meshes, objects, cam)
    r = Renderer(0)
    up = r.upload_scene_built_on_device(meshes, objects, ref.materials, ref.lights)
    for b in (S.BIND_INSTANCES, S.BIND_TLAS_NODES, S.BIND_TLAS_INDICES, S.BIND_TRIANGLES):
        assert up[b].tobytes() == ref.arrays[b].tobytes(), b
    W, H, spp, bn = 96, 54, 3, 5
    r.set_frame(frame_params(cam, W, H, len(ref.lights), bn, spp))
    r.render()                                                   # re-layout straight from the device-resident arrays
    got = r.read_accum()
    want = oracle_render(ref, W, H, spp, bn)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # the arrays that never visited the host are the host builder's, byte for byte
    assert r.read_binding(S.BIND_BLAS_NODES).tobytes() == ref.arrays[S.BIND_BLAS_NODES].tobytes()
    assert r.read_binding(S.BIND_BLAS_INDICES).tobytes() == ref.arrays[S.BIND_BLAS_INDICES].tobytes()
    # a glBufferSubData-style patch of a device-resident array still works (host copy fetched first), and so does
    # replacing the geometry by a plain upload afterwards
    nodes = r.read_binding(S.BIND_BLAS_NODES)
    r.update(S.BIND_BLAS_NODES, nodes[:4])
    r.render()
    assert (r.read_accum().view(np.uint32) == want.view(np.uint32)).all()
    r.upload_scene(ref)
    r.render()
    assert (r.read_accum().view(np.uint32) == want.view(np.uint32)).all()
    r.close()


def test_build_geometry_argument_checks():
    r = Renderer(0)
    tris = S.make_cube(0)
    with pytest.raises(RayZenError) as e:
        r.build_geometry(tris, [(6, 12)])                        # range past the end
    assert e.value.code == -1
    assert r.build_geometry(np.zeros(0, S.TRIANGLE), [(0, 0)])[0]["n_nodes"] == 1      # one empty mesh: an inverted root
    r.close()


def test_a_failed_build_leaves_the_previous_geometry_intact():
    """rz_build_geometry used to write mesh after mesh into the context's own node / index arrays, so a call that
    failed half-way (here: std::bad_alloc injected at the second mesh by the allocation hook) left `geometry on the device`
    describing arrays that now held part of another build -- rz_read_binding returned garbage without an error.  It builds
    into fresh buffers and swaps on success: after the failure the context reads back and renders the FIRST build."""
    cam = S.Camera(position=(0.0, 5.0, 12.0), target=(0.0, -0.35, -1.0), aspect=16 / 9)
    meshes, objects = _scene_parts(10, 4)
    ref = _host_scene(meshes, objects, cam)
    r = Renderer(0)
    r.upload_scene_built_on_device(meshes, objects, ref.materials, ref.lights)
    W, H, spp, bn = 80, 45, 2, 4
    r.set_frame(frame_params(cam, W, H, len(ref.lights), bn, spp))
    want = oracle_render(ref, W, H, spp, bn)
    r.render()
    assert (r.read_accum().view(np.uint32) == want.view(np.uint32)).all()
    # a different, larger set of meshes; the hook throws at the 2nd allocation site = after the first mesh has been built
    other = [S.make_blob(14, 3.0, 1, seed=3), S.make_blob(9, 1.0, 0, seed=5), S.make_cube(2)]
    tris = np.concatenate(other)
    ranges, first = [], 0
    for m in other:
        ranges.append((first, len(m)))
        first += len(m)
    for nth in (1, 2, 3):
        r.debug_fail_alloc(nth)
        with pytest.raises(RayZenError) as e:
            r.build_geometry(tris, ranges)
        assert e.value.code == -8, e.value
        r.debug_fail_alloc(0)
        assert r.read_binding(S.BIND_BLAS_NODES).tobytes() == ref.arrays[S.BIND_BLAS_NODES].tobytes()
        assert r.read_binding(S.BIND_BLAS_INDICES).tobytes() == ref.arrays[S.BIND_BLAS_INDICES].tobytes()
        assert r.read_binding(S.BIND_TRIANGLES).tobytes() == ref.arrays[S.BIND_TRIANGLES].tobytes()
        r.clear_accum()
        r.render()
        assert (r.read_accum().view(np.uint32) == want.view(np.uint32)).all()
    # and the same call succeeds once the hook is off
    built = r.build_geometry(tris, ranges)
    assert [b["n_nodes"] > 0 for b in built] == [True, True, True]
    r.close()


def test_one_million_triangles_device_resident_timing():
    """Not pass/fail on time: prints what the geometry half of scene assembly costs each way (DESIGN.md quotes it)."""
    import time
    blob, cube = S.make_blob(289, 10.0, 0), S.make_cube(4)
    objects = [(0, S.translate(S.scale(S.identity(), (40.0, 0.5, 40.0)), (0.0, -28.0, 0.0))), (1, S.identity())]
    mats, lights = S.reference_materials(), S.reference_lights()
    cam = S.Camera(position=(0.0, 6.0, 34.0), aspect=16 / 9)
    r = Renderer(0)
    r.upload_scene_built_on_device([cube, blob], objects, mats, lights)      # warm-up (allocations, code load)
    t = time.perf_counter()
    r.upload_scene_built_on_device([cube, blob], objects, mats, lights)
    r.set_frame(frame_params(cam, 64, 36, 2, 2, 1))
    r.render(); r.sync()
    dev = time.perf_counter() - t
    t = time.perf_counter()
    sc = S.stress_scene(n=289, blas_builder=r)                   # rz_build_blas: BLAS on the device, arrays through the host
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, 64, 36, 2, 2, 1))
    r.render(); r.sync()
    via_host = time.perf_counter() - t
    print(f"[geometry] 1 002 264 triangles to first frame: device-resident {dev * 1e3:.0f} ms, via host arrays {via_host * 1e3:.0f} ms "
          f"(both include generating nothing: meshes were ready; the second includes librayzen_host's concatenation)")
    r.close()
