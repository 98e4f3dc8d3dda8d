#!/usr/bin/env python3
"""Generates tests/golden/glref_*.npz: frames rendered by RayZen's OWN fragment shader.

    python tests/golden/make_glref.py          (build container only)

oracle/glref/glref.c loads RayZen/shaders/{vertex,fragment}_shader.glsl from /root/reference at run time and runs them
on the OpenGL implementation this image ships (Mesa 23.2 llvmpipe, OpenGL 4.5 core), with the scene arrays below
uploaded as the shader's SSBOs exactly as RayZen's main.cpp uploads its own.  The fixtures are DATA: the inputs (the
eight SSBO arrays, the camera uniforms, the frame parameters) and the output (FragColor, float32 rgb); no text of the
reference is kept.  They are what pins oracle/rz_oracle.c (tests/test_glref.py), and through it the HIP path.

Scenes: RayZen's own (main.cpp:331-384) with the reference's real meshes/monkey.obj and WITHOUT the `car` object whose
mesh file the reference does not ship (an empty BLAS sends the shader's stack loop out of bounds, FS:426-452: undefined);
BASELINE configs[0]'s Cornell box; the all-materials frame (glass blob, mirror cube); 16 rotated instances; and eight
+ four seeded random scenes of tests/test_fuzz_gpu.py (random materials incl. several glasses, lights, mirrored instances;
budgets 1-2, and 3-8 with up to 8 samples), and a table of llvmpipe's own sin / cos / acos (oracle/glref/probe_math.glsl).

`spp` > 1 renders set FS:676's constant `int numSamples = 1; // increase for better quality` to spp in the text the
harness loads -- the only edit it can make, and the one the product's `spp` stands for.  They are labelled (spp field).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MONKEY = "/root/reference/RayZen/meshes/monkey.obj"


def R(W, H, budget, spp=1, **overlay):
    return dict(W=W, H=H, budget=budget, spp=spp, **overlay)


def scenes():
    from rayzen_amd import scene as S
    from test_fuzz_gpu import random_scene
    yield "rayzen_main", (lambda: S.reference_scene(include_empty=False, monkey_obj=MONKEY)), [
        R(256, 192, 1), R(256, 192, 5),                     # main.cpp:600: the first frames run with budget 1, then 5
        R(160, 120, 2, spp=2),                              # currentIor carried from sample to sample through the glass monkey
        R(160, 120, 3, spp=4),
        R(160, 120, 1, fps=59.9, show_lights=True),
        R(160, 120, 1, fps=7.0, show_bvh=True, bvh_mode=0),
        R(160, 120, 1, fps=142.7, show_bvh=True, bvh_mode=1, selected_blas=2, selected_tri=100)]
    yield "cornell", S.cornell_scene, [R(256, 256, 1, spp=4),            # BASELINE configs[0], exactly
                                      R(128, 128, 2), R(97, 61, 5)]
    yield "bunny24_extras", (lambda: S.bunny_scene(n=24, extras=True)), [R(192, 108, 2, spp=2), R(192, 108, 5)]
    yield "instanced16", (lambda: S.instanced_scene(n=12, count=16)), [R(160, 90, 2), R(160, 90, 5)]
    for seed in (4, 6, 8, 19, 27, 30, 37, 39):
        def make(seed=seed):
            sc, rng = random_scene(1000 + seed)
            W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
            spp, b = int(rng.choice([1, 2, 3])), int(rng.integers(1, 3))
            sc.camera.aspect = W / H
            sc.camera.update()
            sc._fuzz_render = R(W, H, b, spp=spp)
            return sc
        yield f"fuzz{seed}", make, None
    for seed in (9, 19, 33, 37):                             # the same generator, deep budgets: Russian roulette, long specular chains
        def make_deep(seed=seed):
            sc, rng = random_scene(1000 + seed)
            W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
            spp, b = int(rng.choice([1, 2, 3, 8])), int(rng.integers(3, 9))
            sc.camera.aspect = W / H
            sc.camera.update()
            sc._fuzz_render = R(W, H, b, spp=spp)
            return sc
        yield f"fuzzdeep{seed}", make_deep, None


def math_table():
    """Arguments and llvmpipe's own sin / cos / acos for them: what pins math flavour 1 of the oracle (rz_oracle_math.h)."""
    rng = np.random.default_rng(20251005)
    x = np.concatenate([rng.uniform(-20, 20, 3000), rng.uniform(-1e5, 1e5, 3000), rng.uniform(0, 1e8, 3000),
                        rng.uniform(0, 2.5e9, 3000), rng.uniform(0, 1e11, 2000), [0.0, -0.0, 91.2228, 1e30, 3.0e38]]).astype(np.float32)
    y = np.concatenate([rng.uniform(-1, 1, len(x) - 2000 - 6), 1 - 10.0 ** rng.uniform(-7, 0, 2000), [1.0, -1.0, 0.0, 0.5, -0.5, 1.0]]).astype(np.float32)
    y = np.clip(y, -1, 1).astype(np.float32)
    return x, y


def main():
    from rayzen_amd import scene as S
    from oracle.glref import glref
    assert glref.available(), "needs /root/reference and Mesa's swrast_dri.so (the build container)"
    total = 0
    only = set(sys.argv[1:])             # make_glref.py [scene ...]: regenerate these fixtures alone
    for name, make, renders in scenes():
        if only and name not in only:
            continue
        sc = make()
        if renders is None:
            renders = [sc._fuzz_render]
        cam = sc.camera
        data = {f"b{b}": np.frombuffer(np.ascontiguousarray(sc.arrays[b]).tobytes(), np.uint8) for b in S.BINDING_DTYPES}
        data.update(cam_view=cam.view, cam_proj=cam.proj, cam_inv_view=cam.inv_view, cam_inv_proj=cam.inv_proj,
                    cam_pos=np.asarray(cam.position, np.float32))
        info = ""
        for k, r in enumerate(renders):
            kw = {x: r[x] for x in ("fps", "show_lights", "show_bvh", "bvh_mode", "selected_blas", "selected_tri") if x in r}
            img, info = glref.render_scene(sc, r["W"], r["H"], r["budget"], num_samples=r["spp"], **kw)
            assert np.all(img[..., 3] == 1.0)               # FS:821: FragColor = vec4(color, 1.0)
            data[f"out{k}"] = np.ascontiguousarray(img[..., :3])
        data["renders"] = np.array(json.dumps(renders))
        data["gl"] = np.array(info)
        path = os.path.join(HERE, f"glref_{name}.npz")
        np.savez_compressed(path, **data)
        total += os.path.getsize(path)
        print(name, [(r["W"], r["H"], r["budget"], r["spp"]) for r in renders], os.path.getsize(path), "bytes")
    if only and "math_table" not in only:
        return
    x, y = math_table()
    t = glref.probe_math(x, y, 0)
    path = os.path.join(HERE, "glref_math_table.npz")
    np.savez_compressed(path, x=x, y=y, sin=t[:, 0], cos=t[:, 1], acos=t[:, 2], hash=t[:, 3])
    total += os.path.getsize(path)
    print("math table", len(x), os.path.getsize(path), "bytes")
    print("total", total, "bytes")


if __name__ == "__main__":
    main()
