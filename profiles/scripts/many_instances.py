"""Sanity: many instances (TLAS depth ~9-11), per-frame device TLAS rebuild + render."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

for count in (64, 256, 1024):
    side = int(round(count ** 0.5))
    sc = S.instanced_scene(n=12, count=count, aspect=1920 / 1080)
    r = Renderer(0)
    r.upload_scene(sc)
    W, H = 1920, 1080
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), 4, 16))
    r.render(); r.sync(); r.render_history_ms()
    floor = sc.arrays[S.BIND_INSTANCES]["transform"][0].copy()
    t0 = time.perf_counter()
    for f in range(5):
        r.update_transforms(np.stack([floor] + S.instanced_transforms(f + 1, count, spacing=24.0 / side, obj_scale=3.0 / side)))
        r.render()
    r.sync()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"instances": count + 1, "tlas_depth": sc.tlas_depth, "ms_per_frame_wall": round(dt, 2), "kernel_ms": round(float(np.mean(r.render_history_ms())), 2)}), flush=True)
    r.close()
