"""ctypes binding of the CPU oracle (oracle/librz_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (rayzen_amd/) never imports
this module.  Pinned against RayZen's own shader run on Mesa llvmpipe (oracle/glref, tests/test_glref.py);
the host half (BVH / OBJ / flatten) is unpinned beyond node counts -- see oracle/rz_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librz_oracle.so")

# struct layouts restated from RayZen/include/{Mesh,BVH,Material,Light}.h
TRIANGLE = np.dtype([("v0", "<f4", 3), ("pad0", "<f4"), ("v1", "<f4", 3), ("pad1", "<f4"),
                     ("v2", "<f4", 3), ("pad2", "<f4"), ("materialIndex", "<i4"), ("tail", "<i4", 3)])
NODE = np.dtype([("bmin", "<f4", 3), ("leftFirst", "<i4"), ("bmax", "<f4", 3), ("count", "<i4")])
INSTANCE = np.dtype([("blasNodeOffset", "<i4"), ("blasTriOffset", "<i4"), ("meshIndex", "<i4"),
                     ("globalTriOffset", "<i4"), ("transform", "<f4", 16), ("inverseTransform", "<f4", 16)])
MATERIAL = np.dtype([("albedo", "<f4", 3), ("metallic", "<f4"), ("roughness", "<f4"),
                     ("reflectivity", "<f4"), ("transparency", "<f4"), ("ior", "<f4")])
LIGHT = np.dtype([("posdir", "<f4", 4), ("color", "<f4", 3), ("power", "<f4")])
assert (TRIANGLE.itemsize, NODE.itemsize, INSTANCE.itemsize, MATERIAL.itemsize, LIGHT.itemsize) == (64, 32, 144, 32, 32)

COUNTER_FIELDS = ("samples", "traversals", "tlas_nodes", "tlas_leaf_indices", "instances",
                  "blas_nodes", "triangles", "materials", "light_fetches", "pixels",
                  "scatters", "diffuse_scatters", "hemi_draws", "lit_lights", "triangles_past_u")


class _Scene(C.Structure):
    _fields_ = [(n, t) for pair in (
        ("triangles", "n_triangles"), ("materials", "n_materials"), ("lights", "n_lights"),
        ("tlas_nodes", "n_tlas_nodes"), ("tlas_indices", "n_tlas_indices"),
        ("blas_nodes", "n_blas_nodes"), ("blas_indices", "n_blas_indices"),
        ("instances", "n_instances")) for n, t in ((pair[0], C.c_void_p), (pair[1], C.c_size_t))]


class _Frame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("inv_view", C.c_float * 16), ("inv_proj", C.c_float * 16),
                ("cam_pos", C.c_float * 3),
                ("num_lights", C.c_int32), ("bounce_budget", C.c_int32),
                ("spp", C.c_int32), ("sample_base", C.c_int32)]


class _Present(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("view", C.c_float * 16), ("proj", C.c_float * 16),
                ("num_lights", C.c_int32), ("fps", C.c_float), ("show_fps", C.c_int32), ("show_lights", C.c_int32),
                ("show_bvh", C.c_int32), ("bvh_mode", C.c_int32), ("selected_blas", C.c_int32),
                ("selected_tri", C.c_int32)]


class _Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]


def build(force=False):
    """Compile the oracle with its Makefile (gcc only; no GPU needed)."""
    srcs = [os.path.join(_HERE, f) for f in ("rz_oracle.c", "rz_oracle_bvh.c", "rz_oracle_present.c", "rz_oracle.h",
                                            "rz_oracle_math.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.rzo_render.restype = C.c_int
        L.rzo_render.argtypes = [C.POINTER(_Scene), C.POINTER(_Frame), C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Counters)]
        L.rzo_last_threads_busy.restype = C.c_int
        L.rzo_last_threads_busy.argtypes = []
        L.rzo_present.restype = C.c_int
        L.rzo_present.argtypes = [C.POINTER(_Scene), C.POINTER(_Present), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rzo_trace.restype = C.c_int
        L.rzo_trace.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_void_p]
        L.rzo_shadow.restype = C.c_int
        L.rzo_shadow.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_float)]
        for f in ("rzo_sin_f", "rzo_cos_f", "rzo_acos_f"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_float]
        L.rzo_set_math_flavour.restype = None
        L.rzo_set_math_flavour.argtypes = [C.c_int]
        L.rzo_get_math_flavour.restype = C.c_int
        L.rzo_get_math_flavour.argtypes = []
        L.rzo_rand_f.restype = C.c_float
        L.rzo_rand_f.argtypes = [C.c_float, C.c_float]
        L.rzo_hemisphere_f.restype = None
        L.rzo_hemisphere_f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rzo_build_blas.restype = C.c_int
        L.rzo_build_blas.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rzo_build_tlas.restype = C.c_int
        L.rzo_build_tlas.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.rzo_world_bounds.restype = None
        L.rzo_world_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rzo_mat4_inverse.restype = None
        L.rzo_mat4_inverse.argtypes = [C.c_void_p, C.c_void_p]
        L.rzo_load_obj.restype = C.c_int
        L.rzo_load_obj.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
        if os.environ.get("RZO_MATH_FLAVOUR"):         # A/B aid: the process-wide default of rz_oracle_math.h's flavour
            L.rzo_set_math_flavour(int(os.environ["RZO_MATH_FLAVOUR"]))
        _lib = L
    return _lib


def _arr(a, dtype):
    a = np.ascontiguousarray(a)
    if a.dtype.itemsize != np.dtype(dtype).itemsize and a.dtype != np.dtype(dtype):
        raise TypeError(f"expected elements of {np.dtype(dtype).itemsize} bytes, got {a.dtype}")
    return a


class Scene:
    """Holds the eight SSBO arrays (any numpy structured dtype of the right
    element size) alive and exposes the C view."""

    def __init__(self, triangles, materials, lights, tlas_nodes, tlas_indices, blas_nodes, blas_indices, instances):
        self.arrays = dict(
            triangles=_arr(triangles, TRIANGLE), materials=_arr(materials, MATERIAL), lights=_arr(lights, LIGHT),
            tlas_nodes=_arr(tlas_nodes, NODE), tlas_indices=_arr(tlas_indices, np.int32),
            blas_nodes=_arr(blas_nodes, NODE), blas_indices=_arr(blas_indices, np.int32),
            instances=_arr(instances, INSTANCE))
        s = _Scene()
        for k, a in self.arrays.items():
            setattr(s, k, a.ctypes.data if a.size else None)
            setattr(s, "n_" + k, a.shape[0])
        self.c = s


def make_frame(width, height, inv_view, inv_proj, cam_pos, num_lights, bounce_budget, spp, sample_base=0):
    f = _Frame()
    f.width, f.height = int(width), int(height)
    f.inv_view[:] = [float(x) for x in np.asarray(inv_view, np.float32).reshape(16)]
    f.inv_proj[:] = [float(x) for x in np.asarray(inv_proj, np.float32).reshape(16)]
    f.cam_pos[:] = [float(x) for x in np.asarray(cam_pos, np.float32).reshape(3)]
    f.num_lights, f.bounce_budget, f.spp, f.sample_base = int(num_lights), int(bounce_budget), int(spp), int(sample_base)
    return f


def render(scene, frame, accum=None, ior_state=None, crop=None, nthreads=1, want_counters=False):
    """Returns accum (H, W, 4) float32, row 0 = bottom row; optionally counters dict."""
    W, H = frame.width, frame.height
    if accum is None:
        accum = np.zeros((H, W, 4), np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == W * H * 4
    x0, y0, x1, y1 = crop if crop is not None else (0, 0, W, H)
    cnt = _Counters()
    rc = lib().rzo_render(C.byref(scene.c), C.byref(frame), accum.ctypes.data,
                          ior_state.ctypes.data if ior_state is not None else None,
                          x0, y0, x1, y1, int(nthreads), C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"rzo_render failed: {rc}")
    if want_counters:
        return accum, {n: int(getattr(cnt, n)) for n in COUNTER_FIELDS}
    return accum


class math_flavour:
    """with rzo.math_flavour(0): ...  -- which sin / cos / acos the oracle evaluates inside the block (rz_oracle_math.h): 1 = Mesa
    llvmpipe's (the default and the product's), 0 = rounds 1-4's binary64 ones.  tests/helpers.py sets the process default to
    the loaded product library's (rz_math_flavour())."""

    def __init__(self, flavour):
        self.flavour = int(flavour)

    def __enter__(self):
        self.old = lib().rzo_get_math_flavour()
        lib().rzo_set_math_flavour(self.flavour)

    def __exit__(self, *exc):
        lib().rzo_set_math_flavour(self.old)


def last_threads_busy():
    """Threads of the last render() call that rendered at least one pixel."""
    return int(lib().rzo_last_threads_busy())


def present(scene, accum, view, proj, num_lights, fps=0.0, show_fps=True, show_lights=False, show_bvh=False,
            bvh_mode=0, selected_blas=0, selected_tri=0):
    """FS:772-819 on an accumulation buffer (H, W, 4).  Returns (rgb float32 (H,W,3), rgba8 uint8 (H,W,4))."""
    H, W = accum.shape[:2]
    p = _Present()
    p.width, p.height = W, H
    p.view[:] = [float(x) for x in np.asarray(view, np.float32).reshape(16)]
    p.proj[:] = [float(x) for x in np.asarray(proj, np.float32).reshape(16)]
    p.num_lights, p.fps = int(num_lights), float(fps)
    p.show_fps, p.show_lights, p.show_bvh = int(bool(show_fps)), int(bool(show_lights)), int(bool(show_bvh))
    p.bvh_mode, p.selected_blas, p.selected_tri = int(bvh_mode), int(selected_blas), int(selected_tri)
    acc = np.ascontiguousarray(accum, np.float32)
    rgb = np.zeros((H, W, 3), np.float32)
    rgba8 = np.zeros((H, W, 4), np.uint8)
    rc = lib().rzo_present(C.byref(scene.c), C.byref(p), acc.ctypes.data, rgb.ctypes.data, rgba8.ctypes.data)
    if rc != 0:
        raise RuntimeError("rzo_present failed")
    return rgb, rgba8


def algorithmic_bytes(c):
    """bytes the reference shader reads/writes for these counters."""
    return (32 * c["tlas_nodes"] + 4 * c["tlas_leaf_indices"] + 144 * c["instances"] + 32 * c["blas_nodes"]
            + 68 * c["triangles"] + 32 * c["materials"] + 32 * c["light_fetches"] + 16 * c["pixels"])


def trace(scene, origin, direction):
    o = np.asarray(origin, np.float32).copy()
    d = np.asarray(direction, np.float32).copy()
    out = np.zeros(10, np.float32)
    lib().rzo_trace(C.byref(scene.c), o.ctypes.data, d.ctypes.data, out.ctypes.data)
    return dict(hit=bool(out[0]), t=out[1], point=out[2:5].copy(), normal=out[5:8].copy(),
                material=int(out[8]), instance=int(out[9]))


def shadow(scene, origin, direction, max_dist):
    o = np.asarray(origin, np.float32).copy()
    d = np.asarray(direction, np.float32).copy()
    vis = C.c_float(0)
    lit = lib().rzo_shadow(C.byref(scene.c), o.ctypes.data, d.ctypes.data, C.c_float(max_dist), C.byref(vis))
    return bool(lit), float(vis.value)


def build_blas(triangles):
    t = _arr(triangles, TRIANGLE)
    n = t.shape[0]
    nodes = np.zeros(2 * max(n, 1) + 1, NODE)
    idx = np.zeros(max(n, 1), np.int32)
    nn = lib().rzo_build_blas(t.ctypes.data if n else None, n, nodes.ctypes.data, idx.ctypes.data)
    return nodes[:nn].copy(), idx[:n].copy()


def build_tlas(roots):
    r = _arr(roots, NODE)
    n = r.shape[0]
    nodes = np.zeros(2 * max(n, 1), NODE)
    idx = np.zeros(max(n, 1), np.int32)
    ni = C.c_int(0)
    nn = lib().rzo_build_tlas(r.ctypes.data if n else None, n, nodes.ctypes.data, idx.ctypes.data, C.byref(ni))
    return nodes[:nn].copy(), idx[:ni.value].copy()


def world_bounds(root, transform):
    r = np.zeros(1, NODE)
    r[0] = root
    m = np.asarray(transform, np.float32).reshape(16).copy()
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    lib().rzo_world_bounds(r.ctypes.data, m.ctypes.data, mn.ctypes.data, mx.ctypes.data)
    return mn, mx


def mat4_inverse(m):
    """glm::inverse of a column-major mat4 (16 float32)."""
    a = np.asarray(m, np.float32).reshape(16).copy()
    out = np.zeros(16, np.float32)
    lib().rzo_mat4_inverse(a.ctypes.data, out.ctypes.data)
    return out


def load_obj(path, material_index):
    p = os.fsencode(path)
    n = lib().rzo_load_obj(p, material_index, None, 0)
    if n < 0:
        raise FileNotFoundError(path)
    tris = np.zeros(n, TRIANGLE)
    lib().rzo_load_obj(p, material_index, tris.ctypes.data, n)
    return tris
