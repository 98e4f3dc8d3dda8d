"""Seeded random scenes through both implementations: random materials (several transparent ones with different
ior, mirrors, rough metals), point and directional lights (some axis-aligned: 1/0 in the slab test), instances under
rotation / non-uniform scale / mirroring, random cameras, 1..40 spp, 1..8 bounces.  HIP == oracle bit for bit."""
import os

import numpy as np
import pytest

from rayzen_amd import scene as S
from helpers import hip_render, oracle_render, mismatch_report

pytestmark = pytest.mark.gpu


def random_scene(seed, opaque=False):
    rng = np.random.default_rng(seed)
    nm = int(rng.integers(3, 9))
    mats = np.zeros(nm, S.MATERIAL)
    for m in mats:
        m["albedo"] = rng.uniform(0.05, 1.0, 3)
        m["metallic"] = rng.choice([0.0, 0.0, 1.0, rng.uniform()])
        m["roughness"] = rng.choice([0.0, 0.05, 0.3, 1.0, rng.uniform()])
        m["reflectivity"] = rng.choice([0.0, 0.0, 1.0, rng.uniform()])
        m["transparency"] = rng.choice([0.0, 0.0, 0.0, 0.9, 1.0, rng.uniform()])
        m["ior"] = rng.choice([1.0, 1.33, 1.5, 2.4])
        if opaque:
            m["transparency"] = 0.0
    nl = int(rng.integers(0, 4))
    lights = np.zeros(nl, S.LIGHT)
    for l in lights:
        if rng.random() < 0.5:
            l["positionOrDirection"] = (*rng.uniform(-8, 8, 3), 1.0)
            l["power"] = rng.uniform(20, 400)
        else:
            d = rng.choice([np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]), rng.uniform(-1, 1, 3) + 1e-3])
            l["positionOrDirection"] = (*d, 0.0)
            l["power"] = rng.uniform(0.5, 3)
        l["color"] = rng.uniform(0.2, 1.0, 3)
    cam = S.Camera(position=rng.uniform(-1, 1, 3) + (0, 1.5, 9), target=(rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.1), -1.0),
                   fov=float(rng.uniform(40, 90)), aspect=float(rng.uniform(0.7, 2.0)))
    s = S.Scene(materials=mats, lights=lights, camera=cam)
    meshes = []
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 3)
        mat = int(rng.integers(0, nm))
        if kind == 0:
            meshes.append(s.add_mesh(S.make_cube(mat)))
        elif kind == 1:
            meshes.append(s.add_mesh(S.make_blob(int(rng.integers(2, 9)), float(rng.uniform(0.5, 2.0)), mat, seed=int(rng.integers(1, 99)))))
        else:
            t = np.zeros(int(rng.integers(1, 40)), S.TRIANGLE)
            c = rng.uniform(-2, 2, (t.shape[0], 3)).astype(np.float32)
            for k in ("v0", "v1", "v2"):
                t[k] = c + rng.uniform(-0.8, 0.8, (t.shape[0], 3)).astype(np.float32)
            t["materialIndex"] = rng.integers(0, nm, t.shape[0])
            meshes.append(s.add_mesh(t))
    s.add_object(s.add_mesh(S.make_cube(int(rng.integers(0, nm)))), S.translate(S.scale(S.identity(), (9.0, 0.4, 9.0)), (0.0, -6.0, 0.0)))
    for _ in range(int(rng.integers(1, 7))):
        t = S.translate(S.identity(), rng.uniform(-4, 4, 3))
        t = S.rotate(t, float(rng.uniform(0, 6.3)), rng.uniform(-1, 1, 3) + 1e-2)
        t = S.scale(t, rng.choice([1.0, -1.0]) * rng.uniform(0.3, 2.0, 3))
        s.add_object(meshes[int(rng.integers(0, len(meshes)))], t)
    return s.build(share_meshes=bool(rng.integers(0, 2))), rng


N_SMALL = int(os.environ.get("RZ_FUZZ_SEEDS", "64"))      # a one-off soak can ask for more


@pytest.mark.parametrize("seed", range(N_SMALL))
def test_random_scene(seed):
    sc, rng = random_scene(1000 + seed)
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
    spp, b = int(rng.choice([1, 2, 3, 8, 17, 40])), int(rng.integers(1, 9))
    sc.camera.aspect = W / H
    sc.camera.update()
    gpu = hip_render(sc, W, H, spp, b)
    ref = oracle_render(sc, W, H, spp, b, nthreads=16)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), f"seed {seed}: " + mismatch_report(gpu, ref)


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_large_frame_persistent_path(seed):
    sc, rng = random_scene(2000 + seed)
    W, H, spp, b = 1280, 1024, 64, int(rng.integers(2, 7))
    sc.camera.aspect = W / H
    sc.camera.update()
    gpu = hip_render(sc, W, H, spp, b)
    from helpers import oracle_frame, oracle_scene
    from oracle import rzo
    ref = np.zeros_like(gpu)
    for y0 in (96, 504, 896):
        rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, spp, b), accum=ref, crop=(0, y0, W, y0 + 8), nthreads=16)
        assert (gpu[y0:y0 + 8].view(np.uint32) == ref[y0:y0 + 8].view(np.uint32)).all(), \
            f"seed {seed}: " + mismatch_report(gpu[y0:y0 + 8], ref[y0:y0 + 8])


N_CLAIMS = int(os.environ.get("RZ_FUZZ_CLAIM_SEEDS", "96"))      # (a soak of 2 000 on the round-3 build, beside 1 200 of the mixed scenes above: bit-identical)


@pytest.mark.parametrize("seed", range(N_CLAIMS))
def test_random_opaque_scene_through_claims_and_cross_claim_pools(seed, monkeypatch):
    """The round-3 launch machinery on random opaque scenes, forced onto frames small enough for the oracle: persistent
    compacting claims of 2 ... 16 groups (stratified or in runs), at 1 ... 130 spp (several pixels per wave, also when the spp does
    not divide 64; several batches per pixel), parked paths kept in the waves' pools across claims and traced whenever a pool
    holds 24 ... 200 of them (so that pools are traced in the middle of a launch, more than once, and at its end), trace_spread
    for the scenes of three and more instances.  Every pixel against the oracle, bit for bit."""
    sc, rng = random_scene(5000 + seed, opaque=True)
    W, H = int(rng.integers(40, 161)), int(rng.integers(24, 97))
    spp, b = int(rng.choice([1, 2, 3, 8, 16, 17, 40, 64, 100, 130])), int(rng.integers(3, 9))
    sc.camera.aspect = W / H
    sc.camera.update()
    per_claim = int(rng.choice([2, 4, 8, 16]))
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", str(per_claim))
    monkeypatch.setenv("RZ_CLAIM_RUN", str(int(rng.choice([1, 2, per_claim]))))
    monkeypatch.setenv("RZ_CROSS_CLAIM_POOL", "1")
    monkeypatch.setenv("RZ_WPOOL_CHUNK", str(int(rng.choice([24, 64, 200]))))
    from rayzen_amd.renderer import Renderer
    r = Renderer(0)
    gpu = hip_render(sc, W, H, spp, b, renderer=r)
    plan = r.debug_last_plan()
    name = r.last_kernel_name()
    r.close()
    assert plan["per_claim"] > 0 and plan["claim_units"] in (8, 16), plan        # the launch did take compacting claims
    assert name == "rz_render_samples+pool", name
    ref = oracle_render(sc, W, H, spp, b, nthreads=16)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), f"seed {seed} ({W}x{H}, {spp} spp, {b} bounces, claims of {per_claim}): " + mismatch_report(gpu, ref)


@pytest.mark.parametrize("seed", range(4))
def test_a_launch_that_gets_no_memory_for_its_cross_claim_pools_renders_the_same_frame(seed, monkeypatch):
    """The pools' scratch (1.5 KB per unit of the launch) is optional: a launch that cannot have it -- here: is told so -- lets
    every claim work its parked paths off by itself, and the frame is the same."""
    sc, rng = random_scene(7000 + seed, opaque=True)
    W, H, spp, b = 128, 72, int(rng.choice([16, 64, 100])), 5
    sc.camera.aspect = W / H
    sc.camera.update()
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "8")
    monkeypatch.setenv("RZ_CROSS_CLAIM_POOL", "1")
    from rayzen_amd.renderer import Renderer
    frames, names = [], []
    for no_memory in ("0", "1"):
        monkeypatch.setenv("RZ_DEBUG_NO_POOL_MEMORY", no_memory)
        r = Renderer(0)
        frames.append(hip_render(sc, W, H, spp, b, renderer=r))
        names.append(r.last_kernel_name())
        r.close()
    assert names == ["rz_render_samples+pool", "rz_render_samples"], names
    ref = oracle_render(sc, W, H, spp, b, nthreads=16)
    for f in frames:
        assert (f.view(np.uint32) == ref.view(np.uint32)).all(), f"seed {seed}: " + mismatch_report(f, ref)


# ---- transparent scenes on the compacting claims (round 4: glass_resolve_unit, the redo list, may_hit_glass) ------------------

N_GLASS_CLAIMS = int(os.environ.get("RZ_FUZZ_GLASS_CLAIM_SEEDS", "96"))       # (a soak of 20 000 on round 4's final build, with 5 000 of the opaque claims and 3 000 of the mixed scenes: bit-identical)


def _transparent_scene(seed):
    """A random scene in which at least one instanced mesh uses a transparent material (so that the launch is a transparent one)."""
    for k in range(50):
        sc, rng = random_scene(seed + 100000 * k)
        tri = sc.arrays[S.BIND_TRIANGLES]
        if (sc.materials["transparency"][tri["materialIndex"]] > 0).any():
            return sc, rng
    raise AssertionError("no transparent scene found")


@pytest.mark.parametrize("seed", range(N_GLASS_CLAIMS))
def test_random_transparent_scene_through_claims(seed, monkeypatch):
    """FS:674's currentIor on the compacting claims, forced onto frames small enough for the oracle: random scenes with several
    transparent materials of different ior (so that a sample can need more than two versions), mirrors and diffuse surfaces,
    1 ... 130 spp (several pixels per wave, also when the spp does not divide 64; several batches per pixel, currentIor carried
    from batch to batch), claims of 2 ... 16 groups, stratified or in runs, pools traced in the middle of a launch and at its
    end, few wait slots (claims that must wait for the pool), frames continued with sample_base > 0 (currentIor carried from
    launch to launch).  Units in which a sample reads currentIor are resolved in the wave from snapshots; pooled paths that
    meet glass send their group to the redo list.  Every pixel against the oracle, bit for bit.
    Reference semantics: fragment_shader.glsl:674, 723-746."""
    sc, rng = _transparent_scene(9000 + seed)
    W, H = int(rng.integers(40, 161)), int(rng.integers(24, 97))
    spp, b = int(rng.choice([1, 2, 3, 8, 16, 17, 40, 64, 100, 130])), int(rng.integers(2, 9))
    sc.camera.aspect = W / H
    sc.camera.update()
    per_claim = int(rng.choice([2, 4, 8, 16]))
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", str(per_claim))
    monkeypatch.setenv("RZ_CLAIM_RUN", str(int(rng.choice([1, 2, per_claim]))))
    monkeypatch.setenv("RZ_WPOOL_CHUNK", str(int(rng.choice([24, 64, 200]))))
    if rng.random() < 0.3:
        monkeypatch.setenv("RZ_WAIT_SLOTS", "1")          # (raised to twice the groups of a claim: the fewest a launch may have)
    if rng.random() < 0.25:
        monkeypatch.setenv("RZ_GLASS_BOX_HINT", "0")      # every late path is parked: many groups are rendered again
    chunk = int(rng.choice([0, 0, max(1, spp // 3)]))     # a third of the frames in several launches of `chunk` samples
    from rayzen_amd.renderer import Renderer
    r = Renderer(0)
    gpu = hip_render(sc, W, H, spp, b, renderer=r, chunk=chunk or None)
    plan = r.debug_last_plan()
    name = r.last_kernel_name()
    r.close()
    assert plan["per_claim"] > 0 and plan["claim_units"] in (8, 16) and plan["transparent"] == 1, plan
    assert name == "rz_render_samples<glass>+pool", name
    ref = oracle_render(sc, W, H, spp, b, nthreads=16)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), f"seed {seed} ({W}x{H}, {spp} spp, {b} bounces, claims of {per_claim}, chunk {chunk}): " + mismatch_report(gpu, ref)


@pytest.mark.parametrize("seed", range(8))
def test_transparent_claims_tallies_and_the_group_code(seed, monkeypatch):
    """The counting launch of a transparent scene on claims (every unit resolved in the wave, the tallies those of the chosen
    versions) against the oracle's tallies, and the image of the claims against the speculating group code (RZ_GLASS_CLAIMS=0)."""
    sc, rng = _transparent_scene(9500 + seed)
    W, H, spp, b = 96, 64, int(rng.choice([4, 16, 64, 100])), 5
    sc.camera.aspect = W / H
    sc.camera.update()
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "4")
    img, cnt = hip_render(sc, W, H, spp, b, counted=True)
    ref, rc = oracle_render(sc, W, H, spp, b, nthreads=16, want_counters=True)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(img, ref)
    assert cnt == rc
    monkeypatch.setenv("RZ_GLASS_CLAIMS", "0")
    grp = hip_render(sc, W, H, spp, b)
    assert (grp.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(grp, ref)


@pytest.mark.parametrize("bounces,pooled", [(32767, True), (32768, False), (70000, False)])
def test_a_bounce_budget_beyond_the_pool_entrys_field_does_without_the_pool(bounces, pooled, monkeypatch):
    """A pool entry keeps its path's bounce in 15 bits (rz_kernels.hip: BACK; bit 31 went to the transparent scenes' released
    samples in round 4).  A launch whose bounce budget does not fit runs the plain persistent loop: same frame.  (Russian
    roulette ends every path long before: the oracle finishes in a blink.)"""
    sc, rng = random_scene(4242, opaque=True)
    W, H, spp = 64, 40, 64
    sc.camera.aspect = W / H
    sc.camera.update()
    monkeypatch.setenv("RZ_GROUPS_PER_CLAIM", "4")
    from rayzen_amd.renderer import Renderer
    r = Renderer(0)
    gpu = hip_render(sc, W, H, spp, bounces, renderer=r)
    name = r.last_kernel_name()
    r.close()
    assert name == ("rz_render_samples+pool" if pooled else "rz_render_samples"), name
    ref = oracle_render(sc, W, H, spp, bounces, nthreads=16)
    assert (gpu.view(np.uint32) == ref.view(np.uint32)).all(), mismatch_report(gpu, ref)
