"""Shared helpers for the parity tests: run the oracle and the HIP path on the same scene arrays."""
import numpy as np

from oracle import rzo
from rayzen_amd import scene as S


def sync_oracle_flavour():
    """The oracle evaluates sin / cos / acos as the LOADED product library does (rz_math_flavour(): a compile-time choice of
    rz_device_math.h; oracle/rz_oracle_math.h holds both definitions).  Called once per process by the suite's session fixture,
    __graft_entry__.smoke() and bench.py's oracle legs.  Returns the flavour."""
    global _FLAVOUR
    from rayzen_amd import _lib
    L = _lib.hip()
    _FLAVOUR = int(L.rz_math_flavour()) if hasattr(L, "rz_math_flavour") else 0
    rzo.lib().rzo_set_math_flavour(_FLAVOUR)
    return _FLAVOUR


_FLAVOUR = None


def oracle_scene(scene):
    a = scene.arrays
    return rzo.Scene(a[S.BIND_TRIANGLES], a[S.BIND_MATERIALS], a[S.BIND_LIGHTS], a[S.BIND_TLAS_NODES],
                     a[S.BIND_TLAS_INDICES], a[S.BIND_BLAS_NODES], a[S.BIND_BLAS_INDICES], a[S.BIND_INSTANCES])


def oracle_frame(scene, width, height, spp, bounces, sample_base=0, num_lights=None):
    if _FLAVOUR is None:        # (first oracle frame of a process that did not come through the suite's fixture: a child script)
        sync_oracle_flavour()
    cam = scene.camera
    nl = len(scene.lights) if num_lights is None else num_lights
    return rzo.make_frame(width, height, cam.inv_view, cam.inv_proj, cam.position, nl, bounces, spp, sample_base)


def oracle_render(scene, width, height, spp, bounces, crop=None, nthreads=8, want_counters=False, num_lights=None):
    osc = oracle_scene(scene)
    fr = oracle_frame(scene, width, height, spp, bounces, num_lights=num_lights)
    return rzo.render(osc, fr, crop=crop, nthreads=nthreads, want_counters=want_counters)


BACKENDS = {"auto": 0, "pixel": 1}      # RZ_FLAG_MEGAKERNEL of include/rayzen_hip.h


def hip_render(scene, width, height, spp, bounces, counted=False, chunk=None, tile_rank=0, tile_nranks=1,
               num_lights=None, renderer=None, backend=None):
    import os
    from rayzen_amd.renderer import Renderer, frame_params
    backend = backend or os.environ.get("RZ_TEST_BACKEND")
    r = renderer or Renderer(0, BACKENDS[backend] if backend else 0)
    r.upload_scene(scene)
    nl = len(scene.lights) if num_lights is None else num_lights
    counters = None
    if counted:
        r.set_frame(frame_params(scene.camera, width, height, nl, bounces, spp, 0, tile_rank, tile_nranks))
        counters = r.render_counted()
    else:
        r.render_scene(scene, width, height, spp, bounces, nl, tile_rank, tile_nranks, chunk)
    img = r.read_accum()
    if renderer is None:
        r.close()
    return (img, counters) if counted else img


def linf(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def mismatch_report(gpu, ref, tol=1e-4):
    d = np.abs(gpu.astype(np.float64) - ref.astype(np.float64)).max(axis=-1)
    bad = np.argwhere(d > tol)
    return f"Linf={d.max():.3e}, {len(bad)} pixels > {tol} of {d.size}; first: {bad[:5].tolist()}; " \
           f"bit-identical pixels: {(gpu.view(np.uint32) == ref.view(np.uint32)).all(axis=-1).mean() * 100:.4f}%"
