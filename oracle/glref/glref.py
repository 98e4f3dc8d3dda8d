"""Runs RayZen's own fragment shader on Mesa llvmpipe (oracle/glref/glref.c) -- the reference itself, run here.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: it needs /root/reference (the shaders are read from
there at run time, never copied) and Mesa's swrast_dri.so.  The GPU box has neither; tests there use the fixtures this
module generated (tests/golden/glref_*.npz, made by tests/golden/make_glref.py).
"""
import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(os.path.dirname(_HERE), "_ref")            # oracle/_ref/: git-ignored build outputs
BINARY = os.path.join(REF_DIR, "glref")
SHADER_DIR = "/root/reference/RayZen/shaders"
DRIVER = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"


def available():
    """The reference's shaders and Mesa's software driver are both here (true in the build container only)."""
    return os.path.exists(os.path.join(SHADER_DIR, "fragment_shader.glsl")) and os.path.exists(DRIVER)


_USABLE = None


def usable():
    """available(), the harness builds, and Mesa hands out an OpenGL >= 4.3 core context to it (probed once per process).
    Returns (True, GL string) or (False, reason): the live tests skip with the reason instead of failing on a box whose Mesa differs."""
    global _USABLE
    if _USABLE is None:
        if not available():
            _USABLE = (False, "RayZen's shaders / Mesa's swrast driver not present")
        else:
            try:
                build()
                with tempfile.TemporaryDirectory() as td:
                    blob = os.path.join(td, "probe.blob")
                    eye = np.eye(4, dtype=np.float32).reshape(16)
                    write_blob(blob, {}, eye, eye, eye, eye, np.zeros(3, np.float32), 1, 1, 1, 0)
                    p = subprocess.run([BINARY, blob, os.path.join(td, "o"), SHADER_DIR], env=dict(os.environ, GLREF_DRIVER=DRIVER, GLREF_PROBE="1"),
                                       capture_output=True, text=True, timeout=120)
                _USABLE = (p.returncode == 0, p.stderr.strip()[-300:])
            except Exception as e:          # no gcc, no headers, a driver that will not load ...
                _USABLE = (False, f"{type(e).__name__}: {e}"[:300])
    return _USABLE


def build(force=False):
    src = os.path.join(_HERE, "glref.c")
    if not force and os.path.exists(BINARY) and os.path.getmtime(BINARY) >= os.path.getmtime(src):
        return BINARY
    os.makedirs(REF_DIR, exist_ok=True)
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Wno-unused-parameter", src, "-o", BINARY, "-ldl"])
    return BINARY


def write_blob(path, arrays, view, proj, inv_view, inv_proj, cam_pos, width, height, bounce_budget, num_lights,
               fps=0.0, show_lights=False, show_bvh=False, bvh_mode=0, selected_blas=0, selected_tri=0, num_samples=1):
    """arrays: binding index -> numpy array (the SSBO bytes as RayZen's main.cpp:1072-1119 uploads them)."""
    f32 = lambda a, n: np.ascontiguousarray(a, np.float32).reshape(n).tobytes()
    n_tris = int(arrays[0].shape[0]) if 0 in arrays else 0
    with open(path, "wb") as f:
        f.write(b"RZGL")
        f.write(struct.pack("<12i", 2, int(width), int(height), int(bounce_budget), int(num_lights), n_tris,
                            int(bool(show_lights)), int(bool(show_bvh)), int(bvh_mode), int(selected_blas), int(selected_tri),
                            int(num_samples)))
        f.write(struct.pack("<f", float(fps)))
        f.write(f32(view, 16) + f32(proj, 16) + f32(inv_view, 16) + f32(inv_proj, 16) + f32(cam_pos, 3))
        for b in range(10):
            a = arrays.get(b)
            raw = np.ascontiguousarray(a).tobytes() if a is not None and a.size else b""
            f.write(struct.pack("<Q", len(raw)))
            f.write(raw)


def render(arrays, view, proj, inv_view, inv_proj, cam_pos, width, height, bounce_budget, num_lights, timeout=1800, **kw):
    """FragColor of RayZen's shader for this scene and camera: (H, W, 4) float32, row 0 = the bottom row.
    The shader renders ONE sample per pixel (`numSamples = 1`, FS:676) and draws its FPS digits in the top-left corner.
    num_samples > 1 sets that one constant in the text the harness loads (glref.c) -- the product's `spp`."""
    build()
    with tempfile.TemporaryDirectory() as td:
        blob, out = os.path.join(td, "scene.blob"), os.path.join(td, "out.f32")
        write_blob(blob, arrays, view, proj, inv_view, inv_proj, cam_pos, width, height, bounce_budget, num_lights, **kw)
        env = dict(os.environ, GLREF_DRIVER=DRIVER)
        p = subprocess.run([BINARY, blob, out, SHADER_DIR], env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError(f"glref failed ({p.returncode}): {p.stderr[-2000:]}")
        img = np.fromfile(out, np.float32).reshape(height, width, 4)
    return img, p.stderr.strip().splitlines()[0] if p.stderr.strip() else ""


def render_scene(scene, width, height, bounce_budget, num_lights=None, **kw):
    """`scene` is a rayzen_amd.scene.Scene (arrays + camera)."""
    cam = scene.camera
    nl = len(scene.lights) if num_lights is None else num_lights
    return render(scene.arrays, cam.view, cam.proj, cam.inv_view, cam.inv_proj, cam.position, width, height, bounce_budget,
                  nl, **kw)


def probe_math(x, y, mode):
    """Tables of llvmpipe's OWN built-ins (oracle/glref/probe_math.glsl, not RayZen's shader): for each pair (x[i], y[i])
    four values -- mode 0: sin(x), cos(x), acos(y), fract(sin(x) * 43758.5453); 1: x / y, sqrt(x), inversesqrt(x), pow(x, y);
    2: dot((x, y), (12.9898, 78.233)), x * y + x, fract(x), length((x, y, 1)); 3: normalize((x, y, 1)), 1 / x.  -> (n, 4) float32."""
    build()
    x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
    n, W = len(x), 1024
    H = (n + W - 1) // W
    buf = np.zeros(2 * W * H, np.float32)
    buf[0:2 * n:2], buf[1:2 * n:2] = x, y
    eye = np.eye(4, dtype=np.float32).reshape(16)
    with tempfile.TemporaryDirectory() as td:
        blob, out = os.path.join(td, "probe.blob"), os.path.join(td, "out.f32")
        write_blob(blob, {0: buf.view(np.uint8)}, eye, eye, eye, eye, np.zeros(3, np.float32), W, H, 1, 0)
        with open(blob, "r+b") as f:            # the header's numTriangles field carries the mode
            f.seek(4 + 4 * 5)
            f.write(struct.pack("<i", int(mode)))
        env = dict(os.environ, GLREF_DRIVER=DRIVER, GLREF_FRAGMENT=os.path.join(_HERE, "probe_math.glsl"))
        p = subprocess.run([BINARY, blob, out, SHADER_DIR], env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError(f"glref probe failed ({p.returncode}): {p.stderr[-2000:]}")
        return np.fromfile(out, np.float32).reshape(H * W, 4)[:n].copy()
