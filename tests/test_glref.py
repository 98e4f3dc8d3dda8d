"""The pin of the oracle: frames rendered by RayZen's OWN fragment shader (tests/golden/glref_*.npz).

oracle/glref/ runs RayZen/shaders/fragment_shader.glsl -- loaded from /root/reference at run time, in the build container --
on the OpenGL implementation the image ships (Mesa 23.2 llvmpipe, OpenGL 4.5 core); tests/golden/make_glref.py froze its
FragColor for the scenes below together with the SSBO arrays and uniforms it was given.  Here the same inputs go through the
oracle (CPU suite) and through the HIP path behind the C-ABI (GPU suite), and the images are compared.

What can agree, and how closely.  GLSL leaves the precision of sin / cos / acos / pow / inversesqrt to the implementation, and
the shader's hash is `fract(sin(x) * 43758.5453)` with x up to 1e11, where the range reduction decides every bit of sin(x):
any two implementations draw DIFFERENT random numbers from the third path segment on (the first two consume none: FS:696
seeds bounce 0 with (0, 0)).  So:

  (1) MATH FLAVOUR 1 (oracle/rz_oracle_math.h; since round 5 also what the PRODUCT is compiled with: rz_device_math.h,
      rz_math_flavour()) evaluates sin / cos / acos exactly as llvmpipe does -- shown bit for bit against llvmpipe's own tables
      (test_flavour_1_is_llvmpipes_sin_cos_acos) -- and nothing else differently.
      With it the oracle and RayZen's shader draw the same numbers, and EVERY frame, at any budget and sample count, must
      agree pixel by pixel: measured <= 1.3e-5 on RayZen's own scene at its own budget 5, half the pixels bit-identical, a few
      pixels per 10 000 beyond 1e-4 where a comparison sits on a knife edge or a long mirror chain amplifies the last bit.
      This is the test that pins the oracle's reading of the whole path loop: traversal, lighting, shadows through glass,
      refraction / total internal reflection / currentIor, mirror-or-diffuse choice, hemisphere draw, Russian roulette, the
      sum over samples, resolve and overlays.
  (2) FLAVOUR 0 -- rounds 1-4's binary64, correctly rounded built-ins; the product's -DRZ_MATH_FLAVOUR=0 build -- differs from (1)
      in those three functions only.  Against the shader it must agree to rounding wherever no random number is consumed
      (budgets 1-2: every pixel within 1e-4) and, at higher budgets, wherever a path ends within two segments (>= 85 % of the
      pixels); the rest are equally valid samples drawn with another generator.
  The HIP path (GPU suite) is held to (1) or (2) according to the flavour of the library that is loaded -- the default build: (1),
  every frame against the reference's own, pixel by pixel -- and to the oracle in that flavour bit for bit.
"""
import json
import os
import types

import numpy as np
import pytest

from rayzen_amd import scene as S
from oracle import rzo
from helpers import oracle_scene, oracle_frame

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
FIXTURES = sorted(f[len("glref_"):-len(".npz")] for f in os.listdir(GOLDEN)
                  if f.startswith("glref_") and f.endswith(".npz") and f != "glref_math_table.npz")
OVERLAY_KEYS = ("fps", "show_lights", "show_bvh", "bvh_mode", "selected_blas", "selected_tri")


def load(name):
    """-> (scene shim with .arrays / .camera / .lights, [render dict], [FragColor rgb (H, W, 3)], GL string)"""
    z = np.load(os.path.join(GOLDEN, f"glref_{name}.npz"))
    arrays = {b: np.frombuffer(z[f"b{b}"].tobytes(), dt).copy() for b, dt in S.BINDING_DTYPES.items()}
    cam = types.SimpleNamespace(view=z["cam_view"], proj=z["cam_proj"], inv_view=z["cam_inv_view"], inv_proj=z["cam_inv_proj"],
                                position=z["cam_pos"])
    sc = types.SimpleNamespace(arrays=arrays, camera=cam, lights=arrays[S.BIND_LIGHTS], materials=arrays[S.BIND_MATERIALS])
    renders = json.loads(str(z["renders"]))
    return sc, renders, [z[f"out{k}"] for k in range(len(renders))], str(z["gl"])


def _cases():
    out = []
    for name in FIXTURES:
        z = np.load(os.path.join(GOLDEN, f"glref_{name}.npz"))
        for k, _ in enumerate(json.loads(str(z["renders"]))):
            out.append((name, k))
    return out


CASES = _cases()


def compare(got, want, render, what, same_random_numbers):
    """got: this implementation's FragColor, want: the reference shader's."""
    g, w = got.astype(np.float64), want.astype(np.float64)
    assert np.isfinite(g).all() and np.isfinite(w).all(), what
    d = np.abs(g - w).max(axis=-1)
    bit = float((got.view(np.uint32) == want.view(np.uint32)).all(axis=-1).mean())
    far = float((d > 1e-4).mean())
    msg = (f"{what} {render}: Linf {d.max():.3e}, pixels beyond 1e-4: {far * 100:.3f} %, bit-identical {bit * 100:.1f} %, "
           f"means {g.mean():.6f} vs {w.mean():.6f}")
    assert bit >= 0.25, msg
    if render["budget"] <= 2:
        # (knife edges: one silhouette pixel of fuzz37 sits at 2e-4; on the Cornell box at 256 x 256 one of a pixel's four samples
        #  grazes the cube's edge and hits in one implementation, misses in the other: 0.19 on 1 pixel of 65 536)
        assert (d > 1e-4).sum() <= max(2, 0.002 * d.size), msg
    elif same_random_numbers:
        assert far <= 0.03, msg                              # (fuzzdeep37: 8 samples down six-segment mirror chains, 2.2 %; the others <= 0.02 %)
        assert abs(g.mean() - w.mean()) <= 2e-3 * w.mean(), msg
    else:
        assert far <= 0.15, msg
        assert abs(g.mean() - w.mean()) <= 0.01 * w.mean(), msg
    return msg


def oracle_fragcolor(sc, r, flavour=None):
    """flavour None: the one the suite runs in -- the loaded product library's (conftest: sync_oracle_flavour)."""
    fr = oracle_frame(sc, r["W"], r["H"], r["spp"], r["budget"])
    osc = oracle_scene(sc)
    with rzo.math_flavour(rzo.lib().rzo_get_math_flavour() if flavour is None else flavour):
        acc = rzo.render(osc, fr, nthreads=8)
    rgb, _ = rzo.present(osc, acc, sc.camera.view, sc.camera.proj, len(sc.lights), **{k: r[k] for k in OVERLAY_KEYS if k in r})
    return rgb


def test_fixtures_cover_the_listed_scenes():
    assert {"rayzen_main", "cornell", "bunny24_extras", "instanced16"} <= set(FIXTURES)
    assert sum(f.startswith("fuzz") for f in FIXTURES) >= 12
    _, renders, outs, gl = load("rayzen_main")
    assert "llvmpipe" in gl and "4.5 (Core Profile) Mesa" in gl
    assert [(r["budget"], r["spp"]) for r in renders[:2]] == [(1, 1), (5, 1)]          # main.cpp:600, FS:676
    # RayZen's frame from where RayZen's camera stands: sky above the horizon, not a closed room
    top = outs[1][-40:-20, 100:160]
    assert (top[..., 2] > top[..., 0] + 0.2).all()
    assert any(r["budget"] >= 5 and r["spp"] >= 8 for n in FIXTURES for r in load(n)[1])


def test_flavour_1_is_llvmpipes_sin_cos_acos():
    """14 005 arguments (|x| to 1e11 and beyond 2^31 octants, -0, the hash's own 91.2228; acos up to 1 - 1e-7) whose sin / cos /
    acos llvmpipe itself tabulated (oracle/glref/probe_math.glsl): flavour 1 reproduces every bit; flavour 0 -- correctly
    rounded -- agrees with llvmpipe's sin to an ulp up to |x| ~ 1e5 and is a different function beyond ~1e7."""
    z = np.load(os.path.join(GOLDEN, "glref_math_table.npz"))
    x, y = z["x"], z["y"]
    L = rzo.lib()
    with rzo.math_flavour(1):
        s = np.array([L.rzo_sin_f(float(v)) for v in x], np.float32)
        c = np.array([L.rzo_cos_f(float(v)) for v in x], np.float32)
        a = np.array([L.rzo_acos_f(float(v)) for v in y], np.float32)
        h = np.array([L.rzo_rand_f(float(v) / 12.9898, 0.0) for v in x[:64]], np.float32)     # (flavour applies to FS:188-190 too)
    for got, key in ((s, "sin"), (c, "cos"), (a, "acos")):
        assert (got.view(np.uint32) == z[key].view(np.uint32)).all(), key
    assert np.isfinite(h).all()
    with rzo.math_flavour(0):
        s0 = np.array([L.rzo_sin_f(float(v)) for v in x], np.float32)
    small, big = np.abs(x) < 1e5, np.abs(x) > 1e8
    assert np.abs(s0[small].astype(np.float64) - z["sin"][small]).max() <= 1.2e-7
    assert np.abs(s0[big].astype(np.float64) - z["sin"][big]).max() > 1.0
    assert np.abs(z["acos"].astype(np.float64) - np.arccos(y.astype(np.float64))).max() < 2e-4      # Mesa's polynomial: 1.6e-4


@pytest.mark.parametrize("name,k", CASES)
def test_oracle_with_llvmpipes_built_ins_matches_the_reference_shader_pixel_by_pixel(name, k):
    sc, renders, outs, _ = load(name)
    print(compare(oracle_fragcolor(sc, renders[k], flavour=1), outs[k], renders[k], f"oracle[flavour 1] vs RayZen's shader, {name}[{k}]", True))


@pytest.mark.parametrize("name,k", CASES)
def test_oracle_matches_the_reference_shader(name, k):
    sc, renders, outs, _ = load(name)
    print(compare(oracle_fragcolor(sc, renders[k], flavour=0), outs[k], renders[k], f"oracle[flavour 0] vs RayZen's shader, {name}[{k}]", False))


@pytest.mark.gpu
@pytest.mark.parametrize("name,k", CASES)
def test_hip_matches_the_reference_shader(name, k):
    """The product, through the C-ABI (rz_upload x 8, rz_set_frame, rz_render, rz_present), against RayZen's shader."""
    from rayzen_amd.renderer import Renderer, frame_params
    sc, renders, outs, _ = load(name)
    r = renders[k]
    R = Renderer(0)
    try:
        R.upload_scene(sc)
        R.set_frame(frame_params(sc.camera, r["W"], r["H"], len(sc.lights), r["budget"], r["spp"]))
        R.render()
        R.sync()
        rgb, _ = R.present(**{x: r[x] for x in OVERLAY_KEYS if x in r})
    finally:
        R.close()
    flavour = rzo.lib().rzo_get_math_flavour()          # (the loaded library's: conftest)
    print(compare(rgb, outs[k], r, f"HIP[flavour {flavour}] vs RayZen's shader, {name}[{k}]", flavour == 1))
    ref = oracle_fragcolor(sc, r)
    assert (rgb.view(np.uint32) == ref.view(np.uint32)).all(), "HIP and oracle differ on a fixture scene"


# ---- live: only where the reference and Mesa's software driver are (the build container) ----

def _glref(same_llvmpipe=False):
    from oracle.glref import glref
    ok, why = glref.usable()
    if not ok:
        pytest.skip(f"RayZen's shader cannot be run here (the GPU box has neither the reference nor its Mesa): {why}")
    if same_llvmpipe and why != load("cornell")[3]:
        pytest.skip(f"math flavour 1 replays the built-ins of the fixtures' llvmpipe, this one is another ({why})")
    return glref


@pytest.mark.parametrize("name", ["cornell", "fuzz6", "fuzzdeep33", "rayzen_main"])
def test_the_fixtures_are_what_the_shader_renders_here(name):
    """Same Mesa, same vector width (the GL strings match): the same bits.  Another llvmpipe (a CPU without FMA evaluates its
    sin differently, another Mesa may): the frames must still agree as two implementations of the shader do."""
    glref = _glref()
    sc, renders, outs, gl = load(name)
    for k, r in enumerate(renders[:3]):
        img, info = glref.render_scene(sc, r["W"], r["H"], r["budget"], num_samples=r["spp"], **{x: r[x] for x in OVERLAY_KEYS if x in r})
        got = np.ascontiguousarray(img[..., :3])
        if info == gl:
            assert (got.view(np.uint32) == outs[k].view(np.uint32)).all(), f"{name}[{k}]: llvmpipe renders another frame than the fixture holds"
        elif r["budget"] <= 2:
            compare(got, outs[k], r, f"this llvmpipe ({info}) vs the fixture's ({gl}), {name}[{k}]", False)


def test_the_math_table_is_what_llvmpipe_computes_here():
    glref = _glref()
    _, _, _, gl = load("cornell")
    if glref.usable()[1] != gl:
        pytest.skip(f"another llvmpipe than the fixtures' ({glref.usable()[1]} vs {gl}): its built-ins need not be the tabulated ones")
    z = np.load(os.path.join(GOLDEN, "glref_math_table.npz"))
    t = glref.probe_math(z["x"], z["y"], 0)
    for col, key in enumerate(("sin", "cos", "acos", "hash")):
        assert (t[:, col].view(np.uint32) == z[key].view(np.uint32)).all(), key


@pytest.mark.parametrize("seed", range(100, 116))
def test_random_scenes_against_the_live_shader(seed):
    """Seeded random scenes (tests/test_fuzz_gpu.py's generator: several glasses, mirrors, mirrored / sheared instances,
    0-3 lights) rendered by RayZen's shader NOW and by the oracle: budgets 1-8, 1-8 samples; flavour 1 pixel by pixel,
    flavour 0 where no random number is consumed."""
    glref = _glref(same_llvmpipe=True)
    from test_fuzz_gpu import random_scene
    sc, rng = random_scene(1000 + seed)
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
    spp, b = int(rng.choice([1, 2, 3, 8])), int(rng.integers(1, 9))
    sc.camera.aspect = W / H
    sc.camera.update()
    r = dict(W=W, H=H, budget=b, spp=spp)
    img, _ = glref.render_scene(sc, W, H, b, num_samples=spp)
    want = np.ascontiguousarray(img[..., :3])
    compare(oracle_fragcolor(sc, r, flavour=1), want, r, f"oracle[flavour 1] vs the live shader, seed {seed}", True)
    if b <= 2:
        compare(oracle_fragcolor(sc, r, flavour=0), want, r, f"oracle[flavour 0] vs the live shader, seed {seed}", False)


@pytest.mark.parametrize("config", ["rayzen 800x600", "bench 1920x1080"])
def test_full_size_frames_against_the_live_shader(config):
    """RayZen's own frame at RayZen's own size (main.cpp:27-28: 800 x 600, budget 5, its real monkey.obj) and the benchmark's
    scene at 1080p (BASELINE configs[1]; the shader's one sample per pixel): the pixel coordinates -- and with them the hash's
    arguments -- are those of the full-size frames, which the small fixtures do not reach."""
    glref = _glref(same_llvmpipe=True)
    if config.startswith("rayzen"):
        sc = S.reference_scene(include_empty=False, monkey_obj="/root/reference/RayZen/meshes/monkey.obj")
        r = dict(W=800, H=600, budget=5, spp=1)
    else:
        sc, W, H, _, b = S.named_config("c2")
        r = dict(W=W, H=H, budget=b, spp=1)
    img, _ = glref.render_scene(sc, r["W"], r["H"], r["budget"])
    want = np.ascontiguousarray(img[..., :3])
    got = oracle_fragcolor(sc, r, flavour=1)
    d = np.abs(got.astype(np.float64) - want).max(axis=-1)
    assert (d > 1e-4).sum() <= 1e-5 * d.size + 2, f"{config}: {(d > 1e-4).sum()} pixels of {d.size} beyond 1e-4, Linf {d.max():.3e}"
    assert (got.view(np.uint32) == want.view(np.uint32)).all(axis=-1).mean() > 0.4


@pytest.mark.parametrize("num_lights", [0, 1, 2])
def test_the_num_lights_uniform_against_the_live_shader(num_lights):
    """`uniform int numLights` (FS:100; main.cpp:1368) may be smaller than the light buffer: the first numLights lights count."""
    glref = _glref(same_llvmpipe=True)
    sc = S.bunny_scene(n=24, extras=True)
    img, _ = glref.render_scene(sc, 160, 90, 3, num_lights=num_lights, num_samples=2)
    osc = oracle_scene(sc)
    with rzo.math_flavour(1):
        acc = rzo.render(osc, oracle_frame(sc, 160, 90, 2, 3, num_lights=num_lights), nthreads=8)
    rgb, _ = rzo.present(osc, acc, sc.camera.view, sc.camera.proj, num_lights)
    d = np.abs(img[..., :3].astype(np.float64) - rgb).max(axis=-1)
    assert (d > 1e-4).sum() <= 2, f"numLights {num_lights}: {(d > 1e-4).sum()} pixels beyond 1e-4, Linf {d.max():.3e}"


@pytest.mark.parametrize("which", ["225 instances", "120k triangles"])
def test_deep_trees_against_the_live_shader(which):
    """A TLAS of 226 instances (depth 9: the list walk's skip positions, the shader's stack of TLAS nodes) and a BLAS of 120 000 triangles
    (depth 18: the LDS stack window and its overflow) against RayZen's shader, budget 3, two samples."""
    glref = _glref(same_llvmpipe=True)
    if which.startswith("225"):
        sc = S.instanced_scene(n=6, count=225)
        W, H = 200, 112
    else:
        sc = S.stress_scene(n=100)
        W, H = 160, 90
    r = dict(W=W, H=H, budget=3, spp=2)
    img, _ = glref.render_scene(sc, W, H, 3, num_samples=2)
    compare(oracle_fragcolor(sc, r, flavour=1), np.ascontiguousarray(img[..., :3]), r, f"oracle[flavour 1] vs the live shader, {which}", True)
