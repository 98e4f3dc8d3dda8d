"""Kernel time of the transparent-scene variant (rz_render_samples<glass>) at C2's frame size:
the C2 scene plus a glass blob and a mirror cube, and the C2 scene with the bunny itself made of glass."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

W, H, SPP, B = 1920, 1080, 64, 4
for name, sc in (("extras", S.bunny_scene(n=76, aspect=W / H, extras=True)),
                 ("glass_bunny", S.bunny_scene(n=76, aspect=W / H, bunny_material=3))):
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP))
    r.render(); r.sync(); r.render_history_ms()
    for _ in range(3):
        r.render()
    r.sync()
    ms = sorted(r.render_history_ms())
    print(json.dumps({"scene": name, "kernel": r.last_kernel_name(), "kernel_ms": round(ms[len(ms) // 2], 3)}), flush=True)
    r.close()
