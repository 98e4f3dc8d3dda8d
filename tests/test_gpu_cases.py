"""More GPU parity: edge cases of the C-ABI and the render loop, all through librayzen_hip.so, all against the
CPU oracle on the same arrays (bar: accumulation-buffer Linf < 1e-4; expected: bit-identical)."""
import numpy as np
import pytest

from rayzen_amd import scene as S
from rayzen_amd import dist as D
from helpers import hip_render, oracle_render, linf, mismatch_report

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _eq(gpu, ref):
    assert linf(gpu, ref) < TOL, mismatch_report(gpu, ref)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(gpu, ref)


def test_resolution_not_a_multiple_of_the_tile():
    sc = S.cornell_scene()
    _eq(hip_render(sc, 61, 35, 2, 3), oracle_render(sc, 61, 35, 2, 3))


def test_one_pixel_frame():
    sc = S.cornell_scene()
    _eq(hip_render(sc, 1, 1, 5, 5), oracle_render(sc, 1, 1, 5, 5))


def test_default_bounce_budget_is_five():
    sc = S.cornell_scene()
    a = hip_render(sc, 48, 48, 2, 0)           # uniformBounceBudget <= 0 -> 5 (FS:673)
    _eq(a, oracle_render(sc, 48, 48, 2, 0))
    _eq(a, oracle_render(sc, 48, 48, 2, 5))


@pytest.mark.parametrize("nl", [0, 1, 2, 7])
def test_num_lights_uniform(nl):
    """numLights below, at and above lights.length() (FS:574-575: the loop stops at lights.length())."""
    sc = S.bunny_scene(n=8, extras=True)
    _eq(hip_render(sc, 64, 36, 2, 3, num_lights=nl), oracle_render(sc, 64, 36, 2, 3, num_lights=nl))


def test_chunked_accumulation_is_bit_identical_to_one_call():
    """sample_base > 0 continues a frame: colour sum AND the per-pixel currentIor (FS:674) carry over."""
    sc = S.bunny_scene(n=12, extras=True)      # glass in view: currentIor really changes
    whole = hip_render(sc, 96, 54, 7, 6)
    _eq(whole, oracle_render(sc, 96, 54, 7, 6))
    for chunk in (1, 3):
        part = hip_render(sc, 96, 54, 7, 6, chunk=chunk)
        assert (part.view(np.uint32) == whole.view(np.uint32)).all()


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_tile_sharding_sums_to_the_single_gpu_frame(nranks):
    sc = S.bunny_scene(n=8, extras=True)
    W, H = 100, 52
    full = hip_render(sc, W, H, 2, 4)
    own = D.owner_map(W, H, nranks)
    total = np.zeros_like(full)
    for r in range(nranks):
        part = hip_render(sc, W, H, 2, 4, tile_rank=r, tile_nranks=nranks)
        assert (part[own != r] == 0).all()                     # a rank never touches pixels it does not own
        assert (part[own == r].view(np.uint32) == full[own == r].view(np.uint32)).all()
        total += part                                           # what the RCCL reduce(SUM) does
    assert (total.view(np.uint32) == full.view(np.uint32)).all()


def test_instanced_dynamic_scene_over_frames():
    """C4's shape: shared BLAS, per-frame transforms -> TLAS rebuild on the host -> rz_update of instances+TLAS."""
    from rayzen_amd.renderer import Renderer
    sc = S.instanced_scene(n=8, count=16)
    r = Renderer(0)
    r.upload_scene(sc)
    W, H = 96, 54
    for frame in (0, 1, 5):
        for oid, t in zip(sc.instance_ids, S.instanced_transforms(frame, 16)):
            sc.set_transform(oid, t)
        sc.update_dynamic()
        r.update_dynamic(sc)
        r.render_scene(sc, W, H, 2, 4)
        _eq(r.read_accum(), oracle_render(sc, W, H, 2, 4))
    r.close()


def test_shared_and_duplicated_meshes_render_identically():
    a = S.instanced_scene(n=6, count=4, share_meshes=True)
    b = S.instanced_scene(n=6, count=4, share_meshes=False)
    ga, gb = hip_render(a, 64, 36, 2, 4), hip_render(b, 64, 36, 2, 4)
    assert (ga.view(np.uint32) == gb.view(np.uint32)).all()
    _eq(ga, oracle_render(a, 64, 36, 2, 4))


def test_empty_mesh_instance_and_empty_scene():
    sc = S.Scene()
    e, c = sc.add_mesh(np.zeros(0, S.TRIANGLE)), sc.add_mesh(S.make_cube(1))
    sc.add_object(e)
    sc.add_object(c, S.translate(S.identity(), (0, 0, -4)))
    sc.build()
    _eq(hip_render(sc, 40, 40, 2, 3), oracle_render(sc, 40, 40, 2, 3))
    empty = S.Scene().build()                                   # no objects at all: pure sky
    g = hip_render(empty, 24, 16, 2, 3)
    _eq(g, oracle_render(empty, 24, 16, 2, 3))
    assert (g[..., 3] == 2).all() and (g[..., 2] > 0).all()


def test_mirror_and_glass_heavy_scene_deep_bounces():
    sc = S.bunny_scene(n=10, bunny_material=3, floor_material=2, extras=True)     # glass bunny on a mirror floor
    _eq(hip_render(sc, 80, 45, 3, 8), oracle_render(sc, 80, 45, 3, 8))


def test_resolve_rgba8():
    """FS:772-773: divide by the sample count, clamp to [0,1]; then 8-bit quantisation."""
    from rayzen_amd.renderer import Renderer
    sc = S.cornell_scene()
    r = Renderer(0)
    r.upload_scene(sc)
    r.render_scene(sc, 64, 64, 4, 2)
    acc, rgba = r.read_accum(), r.resolve_rgba8()
    r.close()
    want = np.rint(np.clip(acc[..., :3] / acc[..., 3:4], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert (rgba[..., :3] == want).all() and (rgba[..., 3] == 255).all()


def test_abi_error_paths():
    from rayzen_amd.renderer import RayZenError, Renderer, frame_params
    sc = S.cornell_scene()
    r = Renderer(0)
    with pytest.raises(RayZenError) as e:            # draw before any upload / frame
        r.render()
    assert e.value.code == -5
    r.set_frame(frame_params(sc.camera, 32, 32, 2, 1, 1))
    with pytest.raises(RayZenError) as e:            # bindings missing
        r.render()
    assert e.value.code == -5
    r.upload_scene(sc)
    with pytest.raises(RayZenError) as e:            # glBufferSubData past the end
        r.update(S.BIND_LIGHTS, sc.lights, offset_bytes=32)
    assert e.value.code == -4
    with pytest.raises(RayZenError) as e:            # not a whole number of elements
        r.upload(S.BIND_TRIANGLES, np.zeros(65, np.uint8))
    assert e.value.code == -1
    bad = sc.arrays[S.BIND_INSTANCES].copy()
    bad["blasNodeOffset"][0] = 10 ** 6
    r.upload(S.BIND_TRIANGLES, sc.arrays[S.BIND_TRIANGLES])
    r.upload(S.BIND_INSTANCES, bad)
    with pytest.raises(RayZenError) as e:            # inconsistent scene is rejected on the host, never launched
        r.render()
    assert e.value.code == -6
    badmat = sc.arrays[S.BIND_TRIANGLES].copy()
    badmat["materialIndex"][3] = 99
    r.upload(S.BIND_INSTANCES, sc.arrays[S.BIND_INSTANCES])
    r.upload(S.BIND_TRIANGLES, badmat)
    with pytest.raises(RayZenError) as e:
        r.render()
    assert e.value.code == -6
    r.upload(S.BIND_TRIANGLES, sc.arrays[S.BIND_TRIANGLES])       # and the context recovers
    r.render()
    assert linf(r.read_accum(), oracle_render(sc, 32, 32, 1, 1)) == 0.0
    with pytest.raises(RayZenError):
        r.set_frame(frame_params(sc.camera, 0, 32, 2, 1, 1))
    with pytest.raises(RayZenError):
        r.set_frame(frame_params(sc.camera, 32, 32, 2, 1, 1, tile_rank=2, tile_nranks=2))
    r.close()


def test_full_size_c2_sampled_bands_and_properties():
    """BASELINE's configs[1] at full size (1920x1080, 64 spp, 4 bounces): the oracle cannot run all 132.7 M paths
    in a test, so (a) a sample of full-width bands is compared exactly, (b) size-independent properties hold:
    every pixel got 64 samples, nothing is NaN/negative, the two halves of a 2-rank split sum to the frame."""
    from oracle import rzo
    from helpers import oracle_frame, oracle_scene
    sc = S.bunny_scene(n=76, aspect=1920 / 1080)
    W, H, spp, b = 1920, 1080, 64, 4
    gpu = hip_render(sc, W, H, spp, b)
    assert np.isfinite(gpu).all() and (gpu >= 0).all() and (gpu[..., 3] == spp).all()
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, spp, b)
    ref = np.zeros_like(gpu)
    for y0 in (0, 272, 536, 800, 1072):
        rzo.render(osc, fr, accum=ref, crop=(0, y0, W, y0 + 8), nthreads=16)
        assert (gpu[y0:y0 + 8].view(np.uint32) == ref[y0:y0 + 8].view(np.uint32)).all(), \
            mismatch_report(gpu[y0:y0 + 8], ref[y0:y0 + 8])
    halves = [hip_render(sc, W, H, spp, b, tile_rank=r, tile_nranks=2) for r in (0, 1)]
    assert ((halves[0] + halves[1]).view(np.uint32) == gpu.view(np.uint32)).all()


def test_full_frame_every_pixel_c2_and_a_glass_scene():
    """Every one of the 2 073 600 pixels, not a sample of bands: C2 itself (opaque kernel, 132.7 M paths, about 12 s of
    oracle time on 16 threads) and the glass + mirror variant of the scene at 16 spp / 5 bounces (speculating kernel)."""
    from oracle import rzo
    from helpers import oracle_frame, oracle_scene
    W, H = 1920, 1080
    for sc, spp, b in ((S.bunny_scene(n=76, aspect=W / H), 64, 4), (S.bunny_scene(n=76, aspect=W / H, extras=True), 16, 5)):
        gpu = hip_render(sc, W, H, spp, b)
        ref = rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, spp, b), nthreads=16)
        assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(gpu, ref)


def test_cpp_frontend_example_builds_and_renders(tmp_path):
    """examples/render_scene.cpp: RayZen-style C++ frontend -> Renderer.h -> C-ABI -> GPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, ppm = str(tmp_path / "render_scene"), str(tmp_path / "o.ppm")
    lib = os.path.join(root, "rayzen_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "include"), "-I",
                           os.path.join(root, "rayzen_amd", "csrc", "host"),
                           os.path.join(root, "examples", "render_scene.cpp"), "-L", lib, "-lrayzen_host",
                           "-lrayzen_hip", f"-Wl,-rpath,{lib}", "-o", exe])
    out = subprocess.run([exe, ppm, "96", "54", "2", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    data = open(ppm, "rb").read()
    assert data.startswith(b"P6\n96 54\n255\n") and len(data) == len(b"P6\n96 54\n255\n") + 96 * 54 * 3
    px = np.frombuffer(data[len(b"P6\n96 54\n255\n"):], np.uint8).reshape(54, 96, 3)
    assert px[0].mean() > 60 and px.std() > 10           # sky on top, not a constant image
    # the same frontend with the BLAS built and kept on the GPU (Renderer::initializeSSBOsOnDevice -> rz_build_geometry)
    ppm2 = str(tmp_path / "o2.ppm")
    out = subprocess.run([exe, ppm2, "96", "54", "2", "2", "device"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert open(ppm2, "rb").read() == data


# ---- the one-lane-per-sample path (scenes without transparent triangles) ------------------------------------

def test_sample_kernel_variants_by_scene_transparency():
    from rayzen_amd.renderer import Renderer
    r = Renderer(0)
    opaque = S.bunny_scene(n=6)
    r.upload_scene(opaque)
    r.render_scene(opaque, 32, 24, 2, 3)
    assert r.last_kernel_name() == "rz_render_samples"
    glass = S.bunny_scene(n=6, extras=True)                   # a glass blob is in the scene
    r.upload_scene(glass)
    r.render_scene(glass, 32, 24, 2, 3)
    assert r.last_kernel_name() == "rz_render_samples<glass>"     # speculated currentIor
    # the glass MATERIAL being present in the material array is not enough: a triangle must use it
    assert (opaque.materials["transparency"] > 0).any()
    r.close()


@pytest.mark.parametrize("spp,chunk", [(1, None), (7, None), (64, None), (130, None), (130, 64), (70, 13)])
def test_sample_kernel_spp_shapes_and_continuation(spp, chunk):
    """spp below / at / above a wavefront of samples, and frames continued with sample_base > 0."""
    sc = S.bunny_scene(n=8)
    W, H = 40, 24
    ref = oracle_render(sc, W, H, spp, 4)
    for backend in ("auto", "pixel"):
        _eq(hip_render(sc, W, H, spp, 4, chunk=chunk, backend=backend), ref)


def test_sample_kernel_tile_sharding_and_counters():
    sc = S.bunny_scene(n=8)
    W, H = 72, 40
    ref, rc = oracle_render(sc, W, H, 3, 4, want_counters=True)
    full, gc = hip_render(sc, W, H, 3, 4, counted=True)
    _eq(full, ref)
    assert gc == rc
    total = np.zeros_like(full)
    for r in range(3):
        total += hip_render(sc, W, H, 3, 4, tile_rank=r, tile_nranks=3)
    assert (total.view(np.uint32) == full.view(np.uint32)).all()


def test_c4_shape_instanced_and_c5_shape_deep_tree_small():
    """Scaled-down configs[3] (16 instances, shared BLAS, dynamic TLAS) and configs[4] (bigger mesh, 8 bounces)."""
    c4 = S.instanced_scene(n=12, count=16)
    _eq(hip_render(c4, 96, 54, 4, 4), oracle_render(c4, 96, 54, 4, 4))
    c5 = S.stress_scene(n=40)
    _eq(hip_render(c5, 96, 54, 3, 8), oracle_render(c5, 96, 54, 3, 8))


@pytest.mark.parametrize("bounces,spp", [(1, 8), (2, 70), (3, 5)])
def test_speculated_ior_is_repaired_when_a_sample_ends_inside_glass(bounces, spp):
    """A path that stops inside glass leaves currentIor = 1.5 for the pixel's NEXT sample (FS:674 is outside the sample
    loop).  With a small bounce budget that happens on most glass pixels, so the parallel-sample kernel's speculation
    (incoming ior = what the previous final sample left) is wrong again and again and must be re-run in order."""
    from oracle import rzo
    from helpers import oracle_frame, oracle_scene
    sc = S.bunny_scene(n=10, bunny_material=3, floor_material=0)          # a glass bunny fills the view
    W, H = 64, 36
    ior = np.ones((H, W), np.float32)
    ref = rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, spp, bounces), ior_state=ior, nthreads=8)
    assert (ior != 1.0).sum() > 20                   # the carry really happens in this frame
    for backend in ("auto", "pixel"):
        _eq(hip_render(sc, W, H, spp, bounces, backend=backend), ref)
    # and a frame continued later starts from the carried value
    _eq(hip_render(sc, W, H, spp, bounces, chunk=3), ref)


def test_device_side_tlas_rebuild_is_byte_identical_to_the_host_builder():
    """rz_update_transforms (inverse + world AABBs + BVH.cpp:178-240 on the GPU) vs SceneBuffers::updateDynamic."""
    from rayzen_amd.renderer import Renderer
    sc = S.instanced_scene(n=8, count=16)
    r = Renderer(0)
    r.upload_scene(sc)
    W, H = 96, 54
    for frame in (3, 4, 11):
        xf = [sc.arrays[S.BIND_INSTANCES]["transform"][0].copy()] + S.instanced_transforms(frame, 16)   # floor + 16
        for oid, t in zip(sc.instance_ids, xf[1:]):
            sc.set_transform(oid, t)
        sc.update_dynamic()                                   # host path: the expected arrays
        r.update_transforms(np.stack(xf))                     # device path: only 17 x 64 B cross the bus
        for b in (S.BIND_INSTANCES, S.BIND_TLAS_NODES, S.BIND_TLAS_INDICES):
            assert r.read_binding(b).tobytes() == sc.arrays[b].tobytes(), b
        r.render_scene(sc, W, H, 2, 4)
        _eq(r.read_accum(), oracle_render(sc, W, H, 2, 4))
    # a later glBufferSubData-style patch still works on top of the device-built state
    r.update_dynamic(sc)
    r.render_scene(sc, W, H, 2, 4)
    _eq(r.read_accum(), oracle_render(sc, W, H, 2, 4))
    r.close()


@pytest.mark.parametrize("n", [1, 2, 5, 40, 64, 65, 130, 700, 3000])
def test_device_tlas_matches_host_for_random_transforms(n):
    from rayzen_amd.renderer import Renderer
    rng = np.random.default_rng(n)
    sc = S.Scene()
    mesh = sc.add_mesh(S.make_cube(0))
    ids = [sc.add_object(mesh) for _ in range(n)]
    sc.build(share_meshes=True)
    r = Renderer(0)
    r.upload_scene(sc)
    xf = []
    for i in range(n):
        # every third scene snaps positions to a coarse grid: equal centres, ties in the bounds, one-sided partitions
        pos = rng.uniform(-20, 20, 3) if n % 3 else np.round(rng.uniform(-20, 20, 3) / 8.0) * 8.0
        t = S.translate(S.identity(), pos)
        t = S.rotate(t, float(rng.uniform(0, 6.28)) if n % 3 else 0.0, rng.normal(size=3))
        t = S.scale(t, rng.uniform(0.3, 3.0, 3) if n % 3 else (1.0, 1.0, 1.0))
        xf.append(t)
        sc.set_transform(ids[i], t)
    sc.update_dynamic()
    r.update_transforms(np.stack(xf))
    for b in (S.BIND_INSTANCES, S.BIND_TLAS_NODES, S.BIND_TLAS_INDICES):
        assert r.read_binding(b).tobytes() == sc.arrays[b].tobytes(), b
    r.close()


@pytest.mark.parametrize("kw", [
    dict(fps=142.7), dict(fps=7.25, show_lights=True), dict(fps=999.94, show_bvh=True, bvh_mode=0, show_lights=True),
    dict(fps=60.0, show_bvh=True, bvh_mode=1, selected_blas=2, selected_tri=17),
    dict(fps=0.0, show_fps=False, show_bvh=True, bvh_mode=1, selected_blas=1, selected_tri=10 ** 6),   # not found: no path
])
def test_present_stage_matches_the_shader_tail(kw):
    """FS:772-819 (resolve + BVH wireframe + light markers + FPS digits): HIP kernel vs oracle, float AND 8-bit."""
    from oracle import rzo
    from helpers import oracle_scene
    from rayzen_amd.renderer import Renderer
    sc = S.instanced_scene(n=8, count=4)
    W, H = 200, 112
    r = Renderer(0)
    r.upload_scene(sc)
    r.render_scene(sc, W, H, 2, 3)
    acc = r.read_accum()
    rgb, rgba8 = r.present(**kw)
    want_rgb, want_rgba8 = rzo.present(oracle_scene(sc), acc, sc.camera.view, sc.camera.proj, len(sc.lights), **kw)
    r.close()
    assert (rgb.view(np.uint32) == want_rgb.view(np.uint32)).all(), float(np.abs(rgb - want_rgb).max())
    assert (rgba8 == want_rgba8).all()
    if kw.get("show_fps", True):
        assert (rgba8[H - 24:H - 8, 8:100, :3] == 255).any()          # white digits in the top-left box


@pytest.mark.parametrize("window", ["2", "5"])
def test_blas_stack_overflow_columns_give_the_same_frame(window, monkeypatch):
    """Deep trees keep only a window of the BLAS stack in LDS (so that 16 waves stay on a CU) and the rest in global
    overflow columns.  Force a tiny window on a persistent-size frame: most pushes then go through the overflow."""
    sc = S.bunny_scene(n=24, aspect=1280 / 1024, extras=False)
    W, H, spp, b = 1280, 1024, 64, 3                 # 1.3 M pixel groups: the persistent launch is taken
    ref = hip_render(sc, W, H, spp, b)
    monkeypatch.setenv("RZ_BLAS_STACK_WINDOW", window)
    got = hip_render(sc, W, H, spp, b)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(got, ref)
    y0 = 512
    from helpers import oracle_frame, oracle_scene
    from oracle import rzo
    o = np.zeros_like(got)
    rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, spp, b), accum=o, crop=(0, y0, W, y0 + 8), nthreads=16)
    assert (got[y0:y0 + 8].view(np.uint32) == o[y0:y0 + 8].view(np.uint32)).all()


def test_persistent_launch_glass_scene_chunked_and_windowed(monkeypatch):
    """The persistent-wave launch (large frame, spp >= 64) of the speculating kernel: 128 spp in one call, as two
    continued calls of 64 (currentIor carried through K.ior), and with a 3-entry LDS stack window -- one frame."""
    sc = S.bunny_scene(n=16, aspect=1280 / 1024, extras=True)
    W, H, b = 1280, 1024, 5
    one = hip_render(sc, W, H, 128, b)
    two = hip_render(sc, W, H, 128, b, chunk=64)
    assert (one.view(np.uint32) == two.view(np.uint32)).all(), mismatch_report(two, one)
    monkeypatch.setenv("RZ_BLAS_STACK_WINDOW", "3")
    win = hip_render(sc, W, H, 128, b)
    assert (one.view(np.uint32) == win.view(np.uint32)).all(), mismatch_report(win, one)
    y0 = 400
    from helpers import oracle_frame, oracle_scene
    from oracle import rzo
    o = np.zeros_like(one)
    rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, 128, b), accum=o, crop=(0, y0, W, y0 + 8), nthreads=16)
    assert (one[y0:y0 + 8].view(np.uint32) == o[y0:y0 + 8].view(np.uint32)).all()


@pytest.mark.parametrize("flags", [0, 4])
def test_inverted_and_nan_child_boxes_take_the_generic_slab_test(flags):
    """the octant-specialised slab test picks tmin / tmax by the ray's octant, which equals the shader's per-axis
    min / max (FS:384-385) only for boxes with min <= max.  Caller-uploaded BLAS nodes are not validated for that, so a scene
    with an inverted or NaN child box must keep the generic test -- detected while the pairs are laid out, on the device
    (flags 0) and on the host (RZ_FLAG_HOST_RELAYOUT = 4) -- and rz_render, rz_render_counted and the oracle must agree."""
    from rayzen_amd.renderer import Renderer, frame_params
    sc = S.bunny_scene(n=8)
    nodes = sc.arrays[S.BIND_BLAS_NODES]
    inner = [i for i in range(len(nodes)) if nodes["count"][i] < 0]
    # swap min and max of a few children on one axis, and put a NaN into another child's box
    for k, i in enumerate(inner[3:9]):
        c = nodes["leftFirst"][i] + (k & 1)
        a = k % 3
        lo, hi = nodes["boundsMin"][c][a], nodes["boundsMax"][c][a]
        nodes["boundsMin"][c][a], nodes["boundsMax"][c][a] = hi, lo
    nodes["boundsMax"][nodes["leftFirst"][inner[12]]][1] = np.nan
    W, H, spp, b = 96, 64, 8, 4
    ref = oracle_render(sc, W, H, spp, b)
    r = Renderer(0, flags)
    got = hip_render(sc, W, H, spp, b, renderer=r)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    r.clear_accum()
    r.render_counted()
    counted = r.read_accum()
    r.close()
    _eq(got, ref)
    _eq(counted, ref)


@pytest.mark.gpu
def test_a_kernel_backstop_is_reported_by_rz_sync():
    """the persistent loop's "cannot happen" bounds (a claim without wait slots, a pool that does not drain, currentIor
    chains that do not resolve) used to end in missing pixels with RZ_OK.  A wave that reaches one now sets a bit of the launch's
    error word and rz_sync returns RZ_ERR_INTERNAL once, naming it -- checked through the hook that sets the word as a kernel
    would; a normal frame leaves it clear."""
    from rayzen_amd import _lib
    from rayzen_amd.renderer import Renderer, RayZenError
    L = _lib.hip()
    sc = S.bunny_scene(n=8)
    r = Renderer(0)
    img = hip_render(sc, 64, 40, 64, 4, renderer=r)      # a compacting-capable launch shape, normal: no bits
    r.sync()
    assert L.rz_debug_poke_backstop(r._c, 2 | 4) == 0
    with pytest.raises(RayZenError) as e:
        r.sync()
    assert e.value.code == -9 and "pool did not drain" in str(e.value) and "chains did not resolve" in str(e.value)
    r.sync()                                             # reported once, then clear
    assert (hip_render(sc, 64, 40, 64, 4, renderer=r).view(np.uint32) == img.view(np.uint32)).all()
    r.close()
