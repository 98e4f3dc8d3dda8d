"""BASELINE.json's configurations at FULL size on the GPU, each checked against the CPU oracle.

The oracle cannot run whole frames of these sizes inside a test (C3 alone is 531 M camera paths), so each
configuration is (a) compared bit for bit on a sample of full-width bands / owned tiles rendered by the oracle on
exactly the same arrays, and (b) checked through size-independent properties: every owned pixel received all its
samples, pixels a rank does not own stay zero, nothing is NaN or negative, ranks sum to the single-GPU frame.
Reference semantics: RayZen/shaders/fragment_shader.glsl:668-773 (path loop), :457-503 / :419-454 (TLAS / BLAS)."""
import numpy as np
import pytest

from rayzen_amd import scene as S
from rayzen_amd import dist as D
from helpers import hip_render, oracle_frame, oracle_scene, mismatch_report
from oracle import rzo

pytestmark = pytest.mark.gpu


def _oracle_bands(sc, W, H, spp, bounces, bands, rows=8, nthreads=16):
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, spp, bounces)
    ref = np.zeros((H, W, 4), np.float32)
    for y0 in bands:
        rzo.render(osc, fr, accum=ref, crop=(0, y0, W, y0 + rows), nthreads=nthreads)
    return ref


@pytest.mark.parametrize("rank", [0, 5])
def test_c3_rank_of_eight_at_256_spp(rank):
    """configs[2]: the C2 scene, 1920x1080, 256 spp, tile-sharded over 8 GPUs -- what ONE of the 8 ranks renders.
    256 spp = four 64-sample batches per pixel in the persistent launch."""
    W, H, spp, b, N = 1920, 1080, 256, 4, 8
    sc = S.bunny_scene(n=76, aspect=W / H)
    part = hip_render(sc, W, H, spp, b, tile_rank=rank, tile_nranks=N)
    own = D.owner_map(W, H, N) == rank
    assert np.isfinite(part).all() and (part >= 0).all()
    assert (part[~own] == 0).all()                          # a rank never touches a pixel it does not own
    assert (part[own][:, 3] == spp).all()                   # every owned pixel received all 256 samples
    bands = (264, 536, 808)
    ref = _oracle_bands(sc, W, H, spp, b, bands, rows=8)
    for y0 in bands:
        o = own[y0:y0 + 8]
        g, r = part[y0:y0 + 8][o], ref[y0:y0 + 8][o]
        assert o.sum() == 8 * W // N
        assert (g.view(np.uint32) == r.view(np.uint32)).all(), mismatch_report(g[None], r[None])


def test_c3_all_eight_ranks_sum_to_the_single_gpu_frame_at_reduced_spp():
    """The reduce(SUM) of configs[2] is exact: 8 disjoint tile sets + zeros.  (64 spp keeps the test short; the
    sharding does not depend on spp.)"""
    W, H, spp, b, N = 1920, 1080, 64, 4, 8
    sc = S.bunny_scene(n=76, aspect=W / H)
    from rayzen_amd.renderer import Renderer
    r = Renderer(0)
    full = hip_render(sc, W, H, spp, b, renderer=r)
    total = np.zeros_like(full)
    for k in range(N):
        total += hip_render(sc, W, H, spp, b, tile_rank=k, tile_nranks=N, renderer=r)
    r.close()
    assert (total.view(np.uint32) == full.view(np.uint32)).all()


def test_c4_full_size_sixteen_instances_dynamic():
    """configs[3]: 16 instances of the 69 k-triangle mesh (ONE shared BLAS), 1080p, 16 spp, 4 bounces, transforms
    changing per frame (host TLAS rebuild + rz_update, main.cpp:1138-1207)."""
    from rayzen_amd.renderer import Renderer
    W, H, spp, b = 1920, 1080, 16, 4
    sc = S.instanced_scene(n=76, count=16, aspect=W / H)
    assert sc.arrays[S.BIND_TRIANGLES].shape[0] == 12 + 12 * 76 * 76          # shared: one copy of the mesh
    r = Renderer(0)
    r.upload_scene(sc)
    for frame in (0, 7):
        for oid, t in zip(sc.instance_ids, S.instanced_transforms(frame, 16)):
            sc.set_transform(oid, t)
        sc.update_dynamic()
        r.update_dynamic(sc)
        r.render_scene(sc, W, H, spp, b)
        gpu = r.read_accum()
        plan = r.debug_last_plan()      # 16 spp: 4 pixels per wave, and at 518 400 groups the persistent, compacting grid (round 3)
        assert plan["pixels_per_wave"] == 4 and plan["per_claim"] > 0 and plan["claim_units"] in (8, 16), plan
        assert np.isfinite(gpu).all() and (gpu >= 0).all() and (gpu[..., 3] == spp).all()
        bands = (304, 520, 736)
        ref = _oracle_bands(sc, W, H, spp, b, bands, rows=16)
        for y0 in bands:
            g, o = gpu[y0:y0 + 16], ref[y0:y0 + 16]
            assert (g.view(np.uint32) == o.view(np.uint32)).all(), mismatch_report(g, o)
    r.close()


def _c5_scene(r, W, H):
    sc = S.stress_scene(n=289, aspect=W / H, blas_builder=r)     # BLAS built on the device (same bytes as the host's)
    assert sc.arrays[S.BIND_TRIANGLES].shape[0] == 12 + 12 * 289 * 289
    assert sc.max_blas_depth >= 20
    return sc


def _c5_check(sc, gpu, W, H, spp, b):
    assert np.isfinite(gpu).all() and (gpu >= 0).all() and (gpu[..., 3] == spp).all()
    bands = (600, 1080, 1500)
    ref = _oracle_bands(sc, W, H, spp, b, bands, rows=4)
    for y0 in bands:
        g, o = gpu[y0:y0 + 4], ref[y0:y0 + 4]
        assert (g.view(np.uint32) == o.view(np.uint32)).all(), mismatch_report(g, o)


def test_c5_full_size_one_million_triangles_4k_at_its_stated_128_spp():
    """configs[4] as BASELINE.json states it: 1 002 264 triangles (BLAS depth 21), 3840x2160, 128 spp, 8 bounces -- the
    whole frame on one GPU.  128 spp = two 64-sample batches per pixel, and at 16.6 M (pixel, batch) units the launch
    takes the persistent grid with 16-unit compacting claims of 8 pixels x 2 batches (rz_kernels.hip:
    plan_render_samples) -- a claim shape no smaller test takes; the launch plan is read back and asserted.  The deep
    tree's 20 stack entries fit the LDS window since the TLAS walk keeps no stack: no overflow columns here (the next
    test forces them).  Three 4-row oracle bands at 128 spp are 5.9 M oracle paths.
    Reference semantics: fragment_shader.glsl:668-773."""
    from rayzen_amd.renderer import Renderer
    W, H, spp, b = 3840, 2160, 128, 8
    r = Renderer(0)
    sc = _c5_scene(r, W, H)
    gpu = hip_render(sc, W, H, spp, b, renderer=r)
    assert r.last_kernel_name() in ("rz_render_samples", "rz_render_samples+pool")
    plan = r.debug_last_plan()
    r.close()
    assert plan["batches_per_pixel"] == 2 and plan["pixels_per_wave"] == 1
    assert plan["claim_units"] == 16 and plan["per_claim"] == 8, plan          # 8 pixels x 2 batches per claim
    assert plan["groups"] == W * H and plan["grid"] < plan["groups"] // 64      # persistent: far fewer waves than pixels
    assert plan["overflow_entries"] == 0 and plan["lds_stack_entries"] == sc.max_blas_depth - 1 and plan["transparent"] == 0
    _c5_check(sc, gpu, W, H, spp, b)


def test_c5_full_size_with_the_stack_overflow_columns_forced(monkeypatch):
    """The OVF = true instantiation at full size: with the LDS window of the BLAS stack forced down to 6 entries the
    1 M-triangle mesh keeps the other 14 levels of its traversal stack in the global overflow columns of the resident
    waves (rz_trace.h: push_entry / pop_entry).  64 spp: one batch per pixel, and at 8.3 M units the launch takes 16-unit
    claims of 16 pixels -- same frame size, same tree, bit-identical to the oracle bands."""
    from rayzen_amd.renderer import Renderer
    W, H, spp, b = 3840, 2160, 64, 8
    r = Renderer(0)
    sc = _c5_scene(r, W, H)
    monkeypatch.setenv("RZ_BLAS_STACK_WINDOW", "6")
    gpu = hip_render(sc, W, H, spp, b, renderer=r)
    plan = r.debug_last_plan()
    monkeypatch.delenv("RZ_BLAS_STACK_WINDOW")
    r.close()
    assert plan["lds_stack_entries"] == 6 and plan["overflow_entries"] == sc.max_blas_depth - 1 - 6, plan
    assert plan["per_claim"] == 16 and plan["claim_units"] == 16 and plan["batches_per_pixel"] == 1, plan
    _c5_check(sc, gpu, W, H, spp, b)


def test_nan_ior_on_a_glass_material_terminates_and_matches():
    """a transparent material with ior = NaN made the speculating kernel's version keys never compare equal
    (float ==) and the kernel spin for ever.  The shader just propagates the NaN.  Keys are compared by bit pattern
    now: the frame must come back, NaN exactly where the oracle has NaN, bit-identical elsewhere."""
    sc = S.bunny_scene(n=8, bunny_material=3, floor_material=0)
    sc.materials["ior"][3] = np.nan
    W, H, spp, b = 48, 28, 4, 5
    gpu = hip_render(sc, W, H, spp, b)
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, spp, b)
    ref = rzo.render(osc, fr, nthreads=4)
    gn, rn = np.isnan(gpu), np.isnan(ref)
    assert rn.any()                                             # the NaN really reaches pixels
    assert (gn == rn).all()
    ok = ~rn
    assert (gpu.view(np.uint32)[ok] == ref.view(np.uint32)[ok]).all()


def test_bad_alloc_inside_the_library_becomes_a_status_code():
    """no C++ exception crosses the C-ABI.  The test hook makes the nth host allocation site throw
    std::bad_alloc inside rz_upload / the re-layout reached from rz_render; the call returns RZ_ERR_NO_MEMORY (-8), and
    the context then renders the scene correctly."""
    from rayzen_amd.renderer import RayZenError, Renderer, frame_params
    from helpers import oracle_render
    sc = S.bunny_scene(n=6, extras=True)
    ref = oracle_render(sc, 40, 24, 2, 3)
    # flags 0: the scene is re-laid-out on the device (one host staging vector); 4 = RZ_FLAG_HOST_RELAYOUT: the host
    # re-layout with its growing vectors -- different depths of the same guard
    for flags, depths in ((0, (1,)), (4, (1, 2, 5, 40))):
        r = Renderer(0, flags)
        r.debug_fail_alloc(1)
        with pytest.raises(RayZenError) as e:
            r.upload(S.BIND_TRIANGLES, sc.arrays[S.BIND_TRIANGLES])
        assert e.value.code == -8
        r.upload_scene(sc)
        r.set_frame(frame_params(sc.camera, 40, 24, len(sc.lights), 3, 2))
        for nth in depths:
            r.debug_fail_alloc(nth)
            with pytest.raises(RayZenError) as e:
                r.render()
            assert e.value.code == -8, e.value
        r.debug_fail_alloc(0)
        r.render()
        got = r.read_accum()
        r.close()
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()


# ---- the compacting launch (rz_kernels.hip: render_claim_compact): opaque scene, spp >= 64, persistent grid --------------

def _band_check(sc, W, H, spp, b, got, y0=None, rows=8):
    y0 = (H // 2) // 8 * 8 if y0 is None else y0
    ref = _oracle_bands(sc, W, H, spp, b, (y0,), rows=rows)
    g, o = got[y0:y0 + rows], ref[y0:y0 + rows]
    assert (g.view(np.uint32) == o.view(np.uint32)).all(), mismatch_report(g, o)


@pytest.mark.parametrize("W,H,spp", [(1280, 1024, 64), (1283, 1021, 100), (1280, 1024, 128), (640, 512, 192), (643, 509, 300), (640, 512, 512), (640, 512, 520), (648, 512, 1024), (640, 512, 1100)])
def test_compacting_launch_equals_the_plain_one_and_the_oracle(W, H, spp, monkeypatch):
    """Frame sizes that are / are not multiples of the 8x8 tile; one, two, three, five and eight 64-sample batches per pixel
    (the last one partly filled; a claim is then 4, 2, 1 ... pixels of the 8- or the 16-unit instantiation), nine and sixteen
    batches (16-unit claims of one pixel), eighteen (more than a claim's units: the plain persistent launch); 5 bounces so that paths are parked and re-parked: compaction on == compaction off ==
    oracle band."""
    sc = S.bunny_scene(n=16, aspect=W / H)
    b = 5
    on = hip_render(sc, W, H, spp, b)
    monkeypatch.setenv("RZ_COMPACT", "0")
    off = hip_render(sc, W, H, spp, b)
    monkeypatch.delenv("RZ_COMPACT")
    assert (on.view(np.uint32) == off.view(np.uint32)).all(), mismatch_report(on, off)
    assert (on[..., 3] == spp).all()
    _band_check(sc, W, H, spp, b, on)


def test_compacting_launch_continued_frames_tile_ranks_and_odd_claims(monkeypatch):
    sc = S.bunny_scene(n=16, aspect=1280 / 1024)
    W, H, b = 1280, 1024, 4
    whole = hip_render(sc, W, H, 128, b)
    # 128 spp as two continued calls of 64 (sample_base = 64 reads the accumulation buffer back in the claim's sum pass)
    two = hip_render(sc, W, H, 128, b, chunk=64)
    assert (two.view(np.uint32) == whole.view(np.uint32)).all(), mismatch_report(two, whole)
    # three tile ranks sum to the frame
    total = np.zeros_like(whole)
    for r in range(3):
        total += hip_render(sc, W, H, 128, b, tile_rank=r, tile_nranks=3)
    assert (total.view(np.uint32) == whole.view(np.uint32)).all()
    # claims whose size does not divide the 64 pixels of a tile straddle two tiles (the cached tile coordinates change
    # inside a claim)
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "3")
    odd = hip_render(sc, W, H, 128, b)
    assert (odd.view(np.uint32) == whole.view(np.uint32)).all(), mismatch_report(odd, whole)
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "7")
    odd = hip_render(sc, W, H, 64, b)
    monkeypatch.delenv("RZ_GROUPS_PER_CLAIM")
    ref64 = hip_render(sc, W, H, 64, b)
    assert (odd.view(np.uint32) == ref64.view(np.uint32)).all(), mismatch_report(odd, ref64)
    _band_check(sc, W, H, 128, b, whole)


def test_compacting_launch_counters_equal_the_oracle_s():
    """The instrumented (COUNT) instantiation of the compacting kernel tallies exactly the reference algorithm's touches."""
    sc = S.bunny_scene(n=12, aspect=1280 / 1024)
    W, H, spp, b = 1280, 1024, 64, 4
    img, cnt = hip_render(sc, W, H, spp, b, counted=True)
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, spp, b)
    ref = np.zeros_like(img)
    _, rc = rzo.render(osc, fr, accum=ref, nthreads=16, want_counters=True)
    assert cnt == rc
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


# ---- the compacting launch for FEWER than 64 samples per pixel (round 3): a unit is a group of 64 / spp pixels -----------

@pytest.mark.parametrize("W,H,spp,claim", [(640, 512, 16, 8), (643, 509, 5, 8), (640, 512, 24, 4), (320, 256, 1, 8), (323, 253, 2, 3),
                                           (640, 512, 48, 16), (331, 259, 63, 7), (640, 512, 32, 1)])
def test_compacting_launch_below_64_spp_equals_the_plain_one_and_the_oracle(W, H, spp, claim, monkeypatch):
    """spp < 64: a wave holds 64 / spp pixels (a compact block of the tile) and a claim is `claim` such groups.  Sample
    counts that divide 64 and ones that do not (5 -> 12 pixels per group, which straddle tiles; 24, 48, 63 -> idle lanes),
    frames that are / are not multiples of the tile, claims that do not divide a tile: claims on == one workgroup per
    group == oracle band, and the launch plan says which of the two ran.  (Small frames do not take the persistent grid
    by themselves: RZ_GROUPS_PER_CLAIM forces it, as in the odd-claims test above; C4's full-size test takes it unforced.)
    Reference semantics: fragment_shader.glsl:668-773."""
    from rayzen_amd.renderer import Renderer
    sc = S.bunny_scene(n=16, aspect=W / H)
    b = 5
    plain = hip_render(sc, W, H, spp, b)
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", str(claim))
    r = Renderer(0)
    on = hip_render(sc, W, H, spp, b, renderer=r)
    plan = r.debug_last_plan()
    r.close()
    monkeypatch.delenv("RZ_GROUPS_PER_CLAIM")
    assert plan["per_claim"] == claim and plan["claim_units"] in (8, 16) and plan["pixels_per_wave"] == 64 // spp, plan
    assert (on.view(np.uint32) == plain.view(np.uint32)).all(), mismatch_report(on, plain)
    assert (on[..., 3] == spp).all()
    _band_check(sc, W, H, spp, b, on)


def test_compacting_launch_below_64_spp_continued_frames_tile_ranks_and_counters(monkeypatch):
    sc = S.bunny_scene(n=12, aspect=1280 / 1024)
    W, H, b = 1280, 1024, 4
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "8")
    whole = hip_render(sc, W, H, 48, b)
    # 48 spp as three continued calls of 16 (sample_base > 0 reads the accumulation buffer back in the claim's sum pass)
    three = hip_render(sc, W, H, 48, b, chunk=16)
    # ... which is NOT the same image as one 48-spp call would be only if a sample's arithmetic depended on its lane: it does not
    assert (three.view(np.uint32) == whole.view(np.uint32)).all(), mismatch_report(three, whole)
    total = np.zeros_like(whole)
    for k in range(3):
        total += hip_render(sc, W, H, 48, b, tile_rank=k, tile_nranks=3)
    assert (total.view(np.uint32) == whole.view(np.uint32)).all()
    img, cnt = hip_render(sc, W, H, 16, b, counted=True)
    monkeypatch.delenv("RZ_GROUPS_PER_CLAIM")
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, 16, b)
    ref = np.zeros_like(img)
    _, rc = rzo.render(osc, fr, accum=ref, nthreads=16, want_counters=True)
    assert cnt == rc
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


# ---- the pool a wave keeps ACROSS its claims (round 3: pool_process / pool_trace) -------------------------

@pytest.mark.parametrize("W,H,spp,claim,chunk", [(640, 512, 64, 8, None), (643, 509, 100, 4, None), (640, 512, 16, 8, None), (323, 253, 5, 3, None),
                                                 (640, 512, 128, 8, 3000), (640, 512, 16, 16, 777), (640, 512, 64, 8, 64), (640, 512, 64, 8, 1)])
def test_cross_claim_pool_equals_the_per_claim_pools_and_the_oracle(W, H, spp, claim, chunk, monkeypatch):
    """The waves of a large opaque launch do not work a claim's parked paths off before the next claim: they collect in the
    wave's own pool, third and later segments side by side, and are traced together when `chunk` of them have come together
    (lanes refilling from the list of pending BLAS walks: pool_trace) and at the end of the launch; the claims that wait
    leave their addends behind and their wave replays their ordered sums when its pool has run dry.  Forced here on small frames
    (RZ_CROSS_CLAIM_POOL=1 + claims by RZ_GROUPS_PER_CLAIM) with chunks from 1 (every claim) to 3000 (the end of the launch
    only), sizes that are not multiples of 64, and several pixels per wave: every shape must give the image of the per-claim
    pools, of one workgroup per group, and of the oracle.  6 bounces: paths survive several passes of a pool.
    Reference semantics: fragment_shader.glsl:705-711, 720-769 (the late part of the bounce loop)."""
    from rayzen_amd.renderer import Renderer
    sc = S.bunny_scene(n=16, aspect=W / H)
    b = 6
    plain = hip_render(sc, W, H, spp, b)
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", str(claim))
    monkeypatch.setenv("RZ_CROSS_CLAIM_POOL", "0")
    local = hip_render(sc, W, H, spp, b)
    monkeypatch.setenv("RZ_CROSS_CLAIM_POOL", "1")
    if chunk is not None:
        monkeypatch.setenv("RZ_WPOOL_CHUNK", str(chunk))
    r = Renderer(0)
    cross = hip_render(sc, W, H, spp, b, renderer=r)
    assert r.last_kernel_name() == "rz_render_samples+pool"
    r.close()
    for k in ("RZ_GROUPS_PER_CLAIM", "RZ_CROSS_CLAIM_POOL", "RZ_WPOOL_CHUNK"):
        monkeypatch.delenv(k, raising=False)
    assert (cross.view(np.uint32) == local.view(np.uint32)).all(), mismatch_report(cross, local)
    assert (cross.view(np.uint32) == plain.view(np.uint32)).all(), mismatch_report(cross, plain)
    assert (cross[..., 3] == spp).all()
    _band_check(sc, W, H, spp, b, cross)


def test_cross_claim_pool_when_every_path_survives(monkeypatch):
    """An all-mirror scene: nothing dies before the bounce budget, so a pool that has been traced is as full as before and
    is traced again before the next claim can park behind it (the inner loop of the claim loop)."""
    W, H, spp, b = 640, 512, 64, 7
    sc = S.bunny_scene(n=12, aspect=W / H, bunny_material=2, floor_material=2)
    plain = hip_render(sc, W, H, spp, b)
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "8")
    monkeypatch.setenv("RZ_CROSS_CLAIM_POOL", "1")
    monkeypatch.setenv("RZ_WPOOL_CHUNK", "200")
    cross = hip_render(sc, W, H, spp, b)
    for k in ("RZ_GROUPS_PER_CLAIM", "RZ_CROSS_CLAIM_POOL", "RZ_WPOOL_CHUNK"):
        monkeypatch.delenv(k, raising=False)
    assert (cross.view(np.uint32) == plain.view(np.uint32)).all(), mismatch_report(cross, plain)
    _band_check(sc, W, H, spp, b, cross)


def test_cross_claim_pool_continued_frames_tile_ranks_and_counters(monkeypatch):
    sc = S.bunny_scene(n=12, aspect=1280 / 1024)
    W, H, b = 1280, 1024, 5
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "8")
    monkeypatch.setenv("RZ_CROSS_CLAIM_POOL", "1")
    whole = hip_render(sc, W, H, 128, b)
    two = hip_render(sc, W, H, 128, b, chunk=64)            # sample_base = 64: the deferred sums read the accumulation buffer back
    assert (two.view(np.uint32) == whole.view(np.uint32)).all(), mismatch_report(two, whole)
    total = np.zeros_like(whole)
    for k in range(3):
        total += hip_render(sc, W, H, 128, b, tile_rank=k, tile_nranks=3)
    assert (total.view(np.uint32) == whole.view(np.uint32)).all()
    img, cnt = hip_render(sc, W, H, 64, b, counted=True)
    monkeypatch.delenv("RZ_GROUPS_PER_CLAIM")
    monkeypatch.delenv("RZ_CROSS_CLAIM_POOL")
    osc, fr = oracle_scene(sc), oracle_frame(sc, W, H, 64, b)
    ref = np.zeros_like(img)
    _, rc = rzo.render(osc, fr, accum=ref, nthreads=16, want_counters=True)
    assert cnt == rc
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()
    _band_check(sc, W, H, 128, b, whole)


def test_rayzen_own_scene_full_frame_budget_1_then_5():
    """RayZen's OWN frame (RayZen/src/main.cpp:331-384: camera (0, 0, 3), seven objects incl. the empty `car` mesh and the
    glass monkey; 800 x 600, fragment_shader.glsl:675's one sample per pixel; bounce budget 1 on frame 0 and 5 afterwards,
    main.cpp:600), with the TLAS / instances refreshed per frame as main.cpp:572 does: the WHOLE frame against the oracle,
    both budgets.  The stand-in meshes have Suzanne's extents (rayzen_amd/scene.py: reference_scene), so the camera stands
    outside every object -- round 4's stand-in swallowed it -- and half the frame is sky."""
    from rayzen_amd.renderer import Renderer, frame_params
    sc, W, H, spp, b = S.named_config("ref")
    assert (W, H, spp, b) == (800, 600, 1, 5) and S.camera_clearance(sc) > 0.1
    osc = oracle_scene(sc)
    r = Renderer(0)
    r.upload_scene(sc)
    for budget in (1, 5):
        sc.update_dynamic()
        r.update_dynamic(sc)
        r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), budget, spp))
        cnt = r.render_counted()
        r.render()
        gpu = r.read_accum()
        ref, oc = rzo.render(osc, oracle_frame(sc, W, H, spp, budget), nthreads=16, want_counters=True)
        assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), (budget, mismatch_report(gpu, ref))
        hit = cnt["light_fetches"] / len(sc.lights) / cnt["samples"]
        assert 0.3 < hit < 0.9, hit                      # an open scene: sky above the floor (a closed room reads 1.0)
        for k in ("samples", "traversals", "tlas_nodes", "instances", "blas_nodes", "triangles", "materials", "light_fetches", "scatters"):
            assert cnt[k] == oc[k], (budget, k, cnt[k], oc[k])
    r.close()


def test_rayzen_own_scene_at_16_spp_on_the_claims():
    """The same scene at 16 samples per launch (`ref16`: the transparent scene through claims, pool, late list): full frame."""
    sc, W, H, spp, b = S.named_config("ref16")
    from rayzen_amd.renderer import Renderer
    r = Renderer(0)
    gpu = hip_render(sc, W, H, spp, b, renderer=r)
    plan = r.debug_last_plan()
    r.close()
    assert plan["transparent"] == 1
    ref = rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, spp, b), nthreads=16)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(gpu, ref)
