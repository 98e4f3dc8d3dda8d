"""Python view of the render C-ABI (include/rayzen_hip.h).

`Renderer` is the analogue of what RayZen's main.cpp does around its draw
call: upload the SSBO arrays (main.cpp:1072-1119), refresh the dynamic ones
(main.cpp:1196-1207), send the per-frame uniforms (main.cpp:1356-1379), draw
(main.cpp:637).  Everything it calls lives in librayzen_hip.so; there is no
fallback path.
"""
import ctypes as C

import numpy as np

from . import _lib
from .scene import BINDING_DTYPES, BIND_INSTANCES, BIND_TLAS_INDICES, BIND_TLAS_NODES


class RayZenError(RuntimeError):
    def __init__(self, what, code, message):
        super().__init__(f"{what} failed ({code}): {message}")
        self.code = code


def frame_params(camera, width, height, num_lights, bounce_budget, spp, sample_base=0, tile_rank=0, tile_nranks=1):
    p = _lib.FrameParams()
    p.width, p.height = int(width), int(height)
    p.inv_view[:] = camera.inv_view.tolist()
    p.inv_proj[:] = camera.inv_proj.tolist()
    p.view[:] = camera.view.tolist()
    p.proj[:] = camera.proj.tolist()
    p.cam_pos[:] = camera.position.tolist()
    p.num_lights, p.bounce_budget = int(num_lights), int(bounce_budget)
    p.spp, p.sample_base = int(spp), int(sample_base)
    p.tile_rank, p.tile_nranks = int(tile_rank), int(tile_nranks)
    return p


class Renderer:
    def __init__(self, device=0, flags=0):
        self._L = _lib.hip()
        self._c = self._L.rz_create(int(device), int(flags))
        if not self._c:
            raise RayZenError("rz_create", -2, self._L.rz_last_error(None).decode())
        self.width = self.height = 0

    def close(self):
        c, self._c = getattr(self, "_c", None), None
        if c:
            self._L.rz_destroy(c)

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise RayZenError(what, rc, self._L.rz_last_error(self._c).decode())

    # -- glBufferData / glBufferSubData ------------------------------------
    def upload(self, binding, array):
        a = np.ascontiguousarray(array)
        self._check(self._L.rz_upload(self._c, int(binding), a.ctypes.data if a.nbytes else None, a.nbytes), "rz_upload")

    def update(self, binding, array, offset_bytes=0):
        a = np.ascontiguousarray(array)
        self._check(self._L.rz_update(self._c, int(binding), int(offset_bytes), a.ctypes.data if a.nbytes else None,
                                      a.nbytes), "rz_update")

    def upload_scene(self, scene):
        """initializeSSBOs' eight uploads (main.cpp:1072-1119)."""
        for b in BINDING_DTYPES:
            self.upload(b, scene.arrays[b])

    def update_dynamic(self, scene):
        """The part of updateDynamicBVHAndSSBOs that changes per frame: instances + TLAS."""
        for b in (BIND_INSTANCES, BIND_TLAS_NODES, BIND_TLAS_INDICES):
            self.update(b, scene.arrays[b])

    def update_transforms(self, transforms):
        """Device-side updateDynamicBVHAndSSBOs: transforms = (n, 16) float32, column-major."""
        t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 16)
        self._check(self._L.rz_update_transforms(self._c, t.ctypes.data, t.shape[0]), "rz_update_transforms")

    def build_blas(self, triangles):
        """BVH::buildBLAS (BVH.cpp:99-175, SAH) on the device.  triangles: TRIANGLE_DTYPE array.
        Returns (nodes, indices, depth, device_ms); byte-identical to the reference builder's output."""
        from .scene import BVH_NODE as NODE_DTYPE, TRIANGLE as TRIANGLE_DTYPE
        t = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
        n = t.shape[0]
        nodes = np.zeros(max(2 * n - 1, 1), NODE_DTYPE)
        idx = np.zeros(n, np.int32)
        nn, depth, ms = C.c_size_t(0), C.c_int(0), C.c_float(0)
        self._check(self._L.rz_build_blas(self._c, t.ctypes.data if n else None, n, nodes.ctypes.data, nodes.shape[0],
                                          idx.ctypes.data if n else None, C.byref(nn), C.byref(depth), C.byref(ms)),
                    "rz_build_blas")
        return nodes[:nn.value].copy(), idx, depth.value, ms.value

    def build_geometry(self, triangles, ranges):
        """rz_build_geometry: triangles = all meshes back to back (TRIANGLE dtype), ranges = [(first, count), ...].
        Builds every BLAS on the device, leaves nodes / indices there as bindings 7 / 8 (and the triangles as binding 0).
        Returns one dict per mesh: node_offset, index_offset, n_nodes, depth, root (a BVH_NODE record)."""
        from .scene import BVH_NODE as NODE_DTYPE, TRIANGLE as TRIANGLE_DTYPE
        t = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
        arr = (_lib.MeshBuild * len(ranges))()
        for k, (first, count) in enumerate(ranges):
            arr[k].first_triangle, arr[k].n_triangles = int(first), int(count)
        self._check(self._L.rz_build_geometry(self._c, t.ctypes.data if t.shape[0] else None, t.shape[0], arr, len(ranges)),
                    "rz_build_geometry")
        out = []
        for m in arr:
            root = np.zeros(1, NODE_DTYPE)
            C.memmove(root.ctypes.data, C.addressof(m.root), 32)
            out.append(dict(node_offset=m.node_offset, index_offset=m.index_offset, n_nodes=m.n_nodes, depth=m.depth, root=root[0]))
        return out

    def upload_scene_built_on_device(self, meshes, objects, materials, lights):
        """initializeSSBOs with the geometry half on the device: meshes = list of TRIANGLE arrays, objects = list of
        (mesh index, 16-float column-major transform).  One BLAS per distinct mesh (true instancing).  Instances, world
        boxes and the TLAS are assembled with librayzen_host exactly as SceneBuffers::build does; returns the arrays
        uploaded for bindings 5, 6 and 9 as a dict (nodes / indices stay on the device: read_binding fetches them)."""
        from . import scene as S
        ranges, first = [], 0
        for m in meshes:
            ranges.append((first, len(m)))
            first += len(m)
        tris = np.concatenate([np.ascontiguousarray(m, S.TRIANGLE) for m in meshes]) if meshes else np.zeros(0, S.TRIANGLE)
        built = self.build_geometry(tris, ranges)
        inst = np.zeros(len(objects), S.BVH_INSTANCE)
        roots = np.zeros(len(objects), S.BVH_NODE)
        for i, (mi, xf) in enumerate(objects):
            xf = np.ascontiguousarray(xf, np.float32).reshape(16)
            b = built[mi]
            inst[i] = (b["node_offset"], b["index_offset"], i, ranges[mi][0], xf, S.inverse(xf))
            mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
            r1 = np.zeros(1, S.BVH_NODE)
            r1[0] = b["root"]
            _lib.host().rzh_world_bounds(r1.ctypes.data, xf.ctypes.data, mn.ctypes.data, mx.ctypes.data)
            roots[i] = b["root"]
            roots[i]["boundsMin"], roots[i]["boundsMax"] = mn, mx
        tn, ti = S.build_tlas(roots)
        self.upload(S.BIND_INSTANCES, inst)
        self.upload(S.BIND_TLAS_NODES, tn)
        self.upload(S.BIND_TLAS_INDICES, ti)
        self.upload(S.BIND_MATERIALS, materials)
        self.upload(S.BIND_LIGHTS, lights)
        return {S.BIND_INSTANCES: inst, S.BIND_TLAS_NODES: tn, S.BIND_TLAS_INDICES: ti, S.BIND_TRIANGLES: tris}

    def read_binding(self, binding):
        """The binding's current content in RayZen's layout (what the device built, after update_transforms)."""
        need = C.c_size_t(0)
        self._check(self._L.rz_read_binding(self._c, int(binding), None, 0, C.byref(need)), "rz_read_binding")
        dt = BINDING_DTYPES[int(binding)]
        out = np.zeros(need.value // dt.itemsize, dt)
        self._check(self._L.rz_read_binding(self._c, int(binding), out.ctypes.data if out.nbytes else None, out.nbytes,
                                            C.byref(need)), "rz_read_binding")
        return out

    # -- uniforms + draw -----------------------------------------------------
    def set_frame(self, params):
        self._check(self._L.rz_set_frame(self._c, C.byref(params)), "rz_set_frame")
        self.width, self.height = params.width, params.height

    def set_stream(self, hip_stream):
        self._check(self._L.rz_set_stream(self._c, C.c_void_p(hip_stream)), "rz_set_stream")

    def bind_accum(self, device_ptr, nbytes):
        self._check(self._L.rz_bind_accum(self._c, C.c_void_p(device_ptr), int(nbytes)), "rz_bind_accum")

    def render(self):
        self._check(self._L.rz_render(self._c), "rz_render")

    def render_counted(self):
        cnt = _lib.Counters()
        self._check(self._L.rz_render_counted(self._c, C.byref(cnt)), "rz_render_counted")
        return {n: int(getattr(cnt, n)) for n in _lib.COUNTER_FIELDS}

    def sync(self):
        self._check(self._L.rz_sync(self._c), "rz_sync")

    def clear_accum(self):
        self._check(self._L.rz_clear_accum(self._c), "rz_clear_accum")

    def read_accum(self):
        """(H, W, 4) float32; row 0 = bottom row; rgb = sum of per-sample radiance, a = sample count."""
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._L.rz_read_accum(self._c, out.ctypes.data, out.nbytes), "rz_read_accum")
        return out

    def resolve_rgba8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self._L.rz_resolve_rgba8(self._c, out.ctypes.data, out.nbytes), "rz_resolve_rgba8")
        return out

    def present(self, fps=0.0, show_fps=True, show_lights=False, show_bvh=False, bvh_mode=0, selected_blas=0,
                selected_tri=0):
        """fragment_shader.glsl:772-819: resolve + overlays.  Returns (rgb float32 (H,W,3), rgba8 uint8 (H,W,4))."""
        p = _lib.PresentParams(float(fps), int(bool(show_fps)), int(bool(show_lights)), int(bool(show_bvh)),
                               int(bvh_mode), int(selected_blas), int(selected_tri))
        rgb = np.empty((self.height, self.width, 3), np.float32)
        rgba8 = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self._L.rz_present(self._c, C.byref(p), rgba8.ctypes.data, rgba8.nbytes, rgb.ctypes.data,
                                       rgb.nbytes), "rz_present")
        return rgb, rgba8

    def last_render_ms(self):
        ms, n = C.c_float(0), C.c_int(0)
        self._check(self._L.rz_last_render_ms(self._c, C.byref(ms), C.byref(n)), "rz_last_render_ms")
        return float(ms.value), int(n.value)

    def render_history_ms(self, cap=64):
        """GPU durations (ms) of the render launches issued since the previous call (oldest first)."""
        buf = (C.c_float * cap)()
        n = self._L.rz_render_history_ms(self._c, buf, cap)
        if n < 0:
            self._check(n, "rz_render_history_ms")
        return [float(buf[k]) for k in range(n)]

    def last_kernel_name(self):
        return self._L.rz_last_kernel_name(self._c).decode()

    def debug_fail_alloc(self, nth):
        """Test hook: the nth host allocation site reached from now on throws std::bad_alloc inside the library."""
        self._check(self._L.rz_debug_fail_alloc(self._c, int(nth)), "rz_debug_fail_alloc")

    def debug_last_plan(self):
        """Test hook: how the last render call was launched (rz_launch_plan as a dict)."""
        lp = _lib.LaunchPlan()
        self._check(self._L.rz_debug_last_plan(self._c, C.byref(lp)), "rz_debug_last_plan")
        return {n: int(getattr(lp, n)) for n, _ in _lib.LaunchPlan._fields_}

    def debug_read_layout(self, which):
        """Test hook: the device scene layout as raw bytes (0: DevPair[], 1: DevTri[])."""
        need = C.c_size_t(0)
        self._check(self._L.rz_debug_read_layout(self._c, int(which), None, 0, C.byref(need)), "rz_debug_read_layout")
        out = np.zeros(need.value, np.uint8)
        self._check(self._L.rz_debug_read_layout(self._c, int(which), out.ctypes.data if need.value else None, out.nbytes, C.byref(need)),
                    "rz_debug_read_layout")
        return out

    def accum_device_ptr(self):
        return self._L.rz_accum_device_ptr(self._c)

    # -- convenience -----------------------------------------------------------
    def render_scene(self, scene, width, height, spp, bounce_budget, num_lights=None, tile_rank=0, tile_nranks=1,
                     chunk=None):
        """Upload-free helper: set the frame for `scene.camera` and render spp samples (optionally in
        chunks of `chunk` samples -- bit-identical to one call)."""
        nl = len(scene.lights) if num_lights is None else num_lights
        chunk = spp if not chunk else chunk
        base = 0
        while base < spp:
            k = min(chunk, spp - base)
            self.set_frame(frame_params(scene.camera, width, height, nl, bounce_budget, k, base, tile_rank, tile_nranks))
            self.render()
            base += k


def algorithmic_bytes(c):
    """the bytes RayZen's shader touches for these counts
    (its SSBO element sizes: 32-B node, 4-B TLAS index, 144-B instance, 64-B triangle + 4-B index,
    32-B material, 32-B light, 16-B RGBA32F pixel)."""
    return (32 * c["tlas_nodes"] + 4 * c["tlas_leaf_indices"] + 144 * c["instances"] + 32 * c["blas_nodes"]
            + 68 * c["triangles"] + 32 * c["materials"] + 32 * c["light_fetches"] + 16 * c["pixels"])
