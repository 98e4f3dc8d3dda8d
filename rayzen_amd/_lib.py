"""ctypes loaders for the in-tree shared libraries.

There is no CPU fallback for rendering: if librayzen_hip.so is missing or has
no GPU to talk to, the calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("RAYZEN_HIP_SO") or os.path.join(_HERE, "lib", "librayzen_hip.so")   # override: A/B builds
HOST_SO = os.path.join(_HERE, "lib", "librayzen_host.so")

# the symbols include/rayzen_hip.h declares
HIP_SYMBOLS = (
    "rz_create", "rz_destroy", "rz_last_error", "rz_upload", "rz_update", "rz_update_transforms", "rz_build_blas", "rz_build_geometry", "rz_read_binding",
    "rz_set_frame", "rz_set_stream",
    "rz_bind_accum", "rz_render", "rz_render_counted", "rz_sync", "rz_clear_accum", "rz_read_accum",
    "rz_resolve_rgba8", "rz_present", "rz_last_render_ms", "rz_render_history_ms", "rz_last_kernel_name", "rz_accum_device_ptr", "rz_version", "rz_sizeof",
    "rz_debug_fail_alloc", "rz_debug_read_layout", "rz_debug_last_plan", "rz_source_hash", "rz_stream_handle", "rz_device_count",
    "rz_group_rccl_version", "rz_group_unique_id", "rz_group_create", "rz_group_create_rank", "rz_group_destroy",
    "rz_group_last_error", "rz_group_size", "rz_group_local_count", "rz_group_rank", "rz_group_ctx", "rz_group_upload",
    "rz_group_update", "rz_group_set_frame", "rz_group_render", "rz_group_reduce", "rz_group_sync", "rz_group_read_frame",
    "rz_group_frame_device_ptr", "rz_group_last_reduce_ms", "rz_group_transport", "rz_group_set_transport", "rz_abi_version", "rz_debug_poke_backstop", "rz_math_flavour",
)
ABI_VERSION = 5         # RZ_ABI_VERSION of the include/rayzen_hip.h this file mirrors
# the symbols include/rayzen_host.h declares
HOST_SYMBOLS = (
    "rzh_load_obj", "rzh_build_blas", "rzh_build_tlas", "rzh_world_bounds", "rzh_scene_create",
    "rzh_scene_destroy", "rzh_scene_add_mesh", "rzh_scene_add_object", "rzh_scene_set_transform",
    "rzh_scene_build", "rzh_scene_set_blas_builder", "rzh_scene_update_dynamic", "rzh_scene_buffer", "rzh_scene_depths",
    "rzh_scene_save_cache", "rzh_scene_load_cache", "rzh_scene_build_cached",
    "rzh_camera_matrices", "rzh_mat_translate", "rzh_mat_scale", "rzh_mat_rotate", "rzh_mat_inverse",
    "rzh_make_cube", "rzh_make_blob", "rzh_version",
)


class FrameParams(C.Structure):
    """rz_frame_params of include/rayzen_hip.h."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("inv_view", C.c_float * 16), ("inv_proj", C.c_float * 16),
                ("view", C.c_float * 16), ("proj", C.c_float * 16),
                ("cam_pos", C.c_float * 3),
                ("num_lights", C.c_int32), ("bounce_budget", C.c_int32),
                ("spp", C.c_int32), ("sample_base", C.c_int32),
                ("tile_rank", C.c_int32), ("tile_nranks", C.c_int32)]


COUNTER_FIELDS = ("samples", "traversals", "tlas_nodes", "tlas_leaf_indices", "instances",
                  "blas_nodes", "triangles", "materials", "light_fetches", "pixels",
                  "scatters", "diffuse_scatters", "hemi_draws", "lit_lights", "triangles_past_u")


class PresentParams(C.Structure):
    """rz_present_params of include/rayzen_hip.h."""
    _fields_ = [("fps", C.c_float), ("show_fps", C.c_int32), ("show_lights", C.c_int32), ("show_bvh", C.c_int32),
                ("bvh_mode", C.c_int32), ("selected_blas", C.c_int32), ("selected_tri", C.c_int32)]


class BvhNode(C.Structure):
    """rz_bvh_node of include/rayzen_hip.h."""
    _fields_ = [("boundsMin", C.c_float * 3), ("leftFirst", C.c_int32), ("boundsMax", C.c_float * 3), ("count", C.c_int32)]


class MeshBuild(C.Structure):
    """rz_mesh_build of include/rayzen_hip.h."""
    _fields_ = [("first_triangle", C.c_size_t), ("n_triangles", C.c_size_t), ("node_offset", C.c_int32),
                ("index_offset", C.c_int32), ("n_nodes", C.c_int32), ("depth", C.c_int32), ("root", BvhNode)]


class LaunchPlan(C.Structure):
    """rz_launch_plan of include/rayzen_hip.h."""
    _fields_ = [("groups", C.c_int64), ("grid", C.c_int64), ("per_claim", C.c_int32), ("claim_units", C.c_int32),
                ("batches_per_pixel", C.c_int32), ("pixels_per_wave", C.c_int32), ("lds_stack_entries", C.c_int32),
                ("overflow_entries", C.c_int32), ("transparent", C.c_int32), ("scratch_mib", C.c_int32)]


class Counters(C.Structure):
    """rz_counters of include/rayzen_hip.h."""
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]


_hip = None
_host = None


def hip():
    """librayzen_hip.so with prototypes set.  Raises if it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise RuntimeError(f"{HIP_SO} is missing: run `python -m rayzen_amd.build` (needs hipcc). "
                               "There is no CPU fallback for the render path.")
        L = C.CDLL(HIP_SO)
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        L.rz_create.restype, L.rz_create.argtypes = vp, [i, C.c_uint]
        L.rz_destroy.restype, L.rz_destroy.argtypes = None, [vp]
        L.rz_last_error.restype, L.rz_last_error.argtypes = C.c_char_p, [vp]
        L.rz_upload.restype, L.rz_upload.argtypes = i, [vp, i, vp, sz]
        L.rz_update.restype, L.rz_update.argtypes = i, [vp, i, sz, vp, sz]
        L.rz_update_transforms.restype, L.rz_update_transforms.argtypes = i, [vp, vp, sz]
        L.rz_read_binding.restype, L.rz_read_binding.argtypes = i, [vp, i, vp, sz, C.POINTER(sz)]
        L.rz_build_blas.restype, L.rz_build_blas.argtypes = i, [vp, vp, sz, vp, sz, vp, C.POINTER(sz), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.rz_build_geometry.restype, L.rz_build_geometry.argtypes = i, [vp, vp, sz, C.POINTER(MeshBuild), sz]
        L.rz_set_frame.restype, L.rz_set_frame.argtypes = i, [vp, C.POINTER(FrameParams)]
        L.rz_set_stream.restype, L.rz_set_stream.argtypes = i, [vp, vp]
        L.rz_bind_accum.restype, L.rz_bind_accum.argtypes = i, [vp, vp, sz]
        L.rz_render.restype, L.rz_render.argtypes = i, [vp]
        L.rz_render_counted.restype, L.rz_render_counted.argtypes = i, [vp, C.POINTER(Counters)]
        L.rz_sync.restype, L.rz_sync.argtypes = i, [vp]
        L.rz_clear_accum.restype, L.rz_clear_accum.argtypes = i, [vp]
        L.rz_read_accum.restype, L.rz_read_accum.argtypes = i, [vp, vp, sz]
        L.rz_resolve_rgba8.restype, L.rz_resolve_rgba8.argtypes = i, [vp, vp, sz]
        L.rz_present.restype, L.rz_present.argtypes = i, [vp, C.POINTER(PresentParams), vp, sz, vp, sz]
        L.rz_last_render_ms.restype, L.rz_last_render_ms.argtypes = i, [vp, C.POINTER(C.c_float), C.POINTER(i)]
        L.rz_render_history_ms.restype, L.rz_render_history_ms.argtypes = i, [vp, C.POINTER(C.c_float), i]
        L.rz_last_kernel_name.restype, L.rz_last_kernel_name.argtypes = C.c_char_p, [vp]
        L.rz_accum_device_ptr.restype, L.rz_accum_device_ptr.argtypes = vp, [vp]
        L.rz_version.restype, L.rz_version.argtypes = C.c_char_p, []
        L.rz_sizeof.restype, L.rz_sizeof.argtypes = sz, [i]
        L.rz_debug_fail_alloc.restype, L.rz_debug_fail_alloc.argtypes = i, [vp, i]
        L.rz_debug_read_layout.restype, L.rz_debug_read_layout.argtypes = i, [vp, i, vp, sz, C.POINTER(sz)]
        L.rz_debug_last_plan.restype, L.rz_debug_last_plan.argtypes = i, [vp, C.POINTER(LaunchPlan)]
        L.rz_source_hash.restype, L.rz_source_hash.argtypes = C.c_char_p, []
        L.rz_stream_handle.restype, L.rz_stream_handle.argtypes = vp, [vp]
        L.rz_device_count.restype, L.rz_device_count.argtypes = i, []
        L.rz_group_rccl_version.restype, L.rz_group_rccl_version.argtypes = i, [C.POINTER(i)]
        L.rz_group_unique_id.restype, L.rz_group_unique_id.argtypes = i, [vp]
        L.rz_group_create.restype, L.rz_group_create.argtypes = vp, [i, C.POINTER(i), C.c_uint]
        L.rz_group_create_rank.restype, L.rz_group_create_rank.argtypes = vp, [i, i, i, vp, C.c_uint]
        L.rz_group_destroy.restype, L.rz_group_destroy.argtypes = None, [vp]
        L.rz_group_last_error.restype, L.rz_group_last_error.argtypes = C.c_char_p, [vp]
        L.rz_group_size.restype, L.rz_group_size.argtypes = i, [vp]
        L.rz_group_local_count.restype, L.rz_group_local_count.argtypes = i, [vp]
        L.rz_group_rank.restype, L.rz_group_rank.argtypes = i, [vp, i]
        L.rz_group_ctx.restype, L.rz_group_ctx.argtypes = vp, [vp, i]
        L.rz_group_upload.restype, L.rz_group_upload.argtypes = i, [vp, i, vp, sz]
        L.rz_group_update.restype, L.rz_group_update.argtypes = i, [vp, i, sz, vp, sz]
        L.rz_group_set_frame.restype, L.rz_group_set_frame.argtypes = i, [vp, C.POINTER(FrameParams)]
        L.rz_group_render.restype, L.rz_group_render.argtypes = i, [vp]
        L.rz_group_reduce.restype, L.rz_group_reduce.argtypes = i, [vp, i]
        L.rz_group_sync.restype, L.rz_group_sync.argtypes = i, [vp]
        L.rz_group_read_frame.restype, L.rz_group_read_frame.argtypes = i, [vp, vp, sz]
        L.rz_group_frame_device_ptr.restype, L.rz_group_frame_device_ptr.argtypes = vp, [vp]
        # (each in a try of its own: an A/B library built from an older revision (RAYZEN_HIP_SO) may lack any ONE of them, and
        #  everything else must still work -- ADVICE r4)
        for name, res, args in (("rz_group_last_reduce_ms", i, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
                                ("rz_group_transport", C.c_char_p, [vp]),
                                ("rz_group_set_transport", i, [vp, C.c_char_p]),
                                ("rz_abi_version", i, []), ("rz_debug_poke_backstop", i, [vp, C.c_uint]), ("rz_math_flavour", i, [])):
            try:
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            except AttributeError:
                if not os.environ.get("RAYZEN_HIP_SO"):
                    raise
        # the structs this file mirrors are the ones of ABI revision ABI_VERSION (include/rayzen_hip.h: RZ_ABI_VERSION)
        if hasattr(L, "rz_abi_version") and L.rz_abi_version() != ABI_VERSION and not os.environ.get("RAYZEN_HIP_SO"):
            raise RuntimeError(f"{HIP_SO} speaks ABI revision {L.rz_abi_version()}, this binding {ABI_VERSION}: rebuild (python -m rayzen_amd.build)")
        _hip = L
    return _hip


def host():
    """librayzen_host.so with prototypes set."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise RuntimeError(f"{HOST_SO} is missing: run `python -m rayzen_amd.build`")
        L = C.CDLL(HOST_SO)
        vp, i, sz, f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
        L.rzh_load_obj.restype, L.rzh_load_obj.argtypes = i, [C.c_char_p, i, vp, i]
        L.rzh_build_blas.restype, L.rzh_build_blas.argtypes = i, [vp, i, vp, vp, C.POINTER(i)]
        L.rzh_build_tlas.restype, L.rzh_build_tlas.argtypes = i, [vp, i, vp, vp, C.POINTER(i)]
        L.rzh_world_bounds.restype, L.rzh_world_bounds.argtypes = None, [vp, vp, vp, vp]
        L.rzh_scene_create.restype, L.rzh_scene_create.argtypes = vp, []
        L.rzh_scene_destroy.restype, L.rzh_scene_destroy.argtypes = None, [vp]
        L.rzh_scene_add_mesh.restype, L.rzh_scene_add_mesh.argtypes = i, [vp, vp, i]
        L.rzh_scene_add_object.restype, L.rzh_scene_add_object.argtypes = i, [vp, i, vp]
        L.rzh_scene_set_transform.restype, L.rzh_scene_set_transform.argtypes = i, [vp, i, vp]
        L.rzh_scene_build.restype, L.rzh_scene_build.argtypes = i, [vp, i]
        L.rzh_scene_set_blas_builder.restype, L.rzh_scene_set_blas_builder.argtypes = i, [vp, vp, vp]
        L.rzh_scene_update_dynamic.restype, L.rzh_scene_update_dynamic.argtypes = i, [vp]
        L.rzh_scene_buffer.restype, L.rzh_scene_buffer.argtypes = vp, [vp, i, C.POINTER(sz)]
        L.rzh_scene_save_cache.restype, L.rzh_scene_save_cache.argtypes = i, [vp, C.c_char_p]
        L.rzh_scene_load_cache.restype, L.rzh_scene_load_cache.argtypes = i, [vp, C.c_char_p]
        L.rzh_scene_build_cached.restype, L.rzh_scene_build_cached.argtypes = i, [vp, C.c_char_p, i, C.POINTER(i * 5)]
        L.rzh_scene_depths.restype, L.rzh_scene_depths.argtypes = None, [vp, C.POINTER(i), C.POINTER(i)]
        L.rzh_camera_matrices.restype = None
        L.rzh_camera_matrices.argtypes = [vp, vp, vp, f, f, f, f, vp, vp, vp, vp]
        for n in ("rzh_mat_translate", "rzh_mat_scale"):
            getattr(L, n).restype, getattr(L, n).argtypes = None, [vp, vp, vp]
        L.rzh_mat_rotate.restype, L.rzh_mat_rotate.argtypes = None, [vp, f, vp, vp]
        L.rzh_mat_inverse.restype, L.rzh_mat_inverse.argtypes = None, [vp, vp]
        L.rzh_make_cube.restype, L.rzh_make_cube.argtypes = i, [i, vp, i]
        L.rzh_make_blob.restype, L.rzh_make_blob.argtypes = i, [i, f, C.c_uint, i, vp, i]
        L.rzh_version.restype, L.rzh_version.argtypes = C.c_char_p, []
        _host = L
    return _host
