"""What does the transparent-scene variant cost when no path ever meets glass?  C2 plus ONE glass triangle behind the
camera (forces rz_render_samples<glass>), against plain C2."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

W, H, SPP, B = 1920, 1080, 64, 4


def scene(with_glass):
    s = S.Scene(camera=S.Camera(position=(0.0, 2.5, 10.0), aspect=W / H))
    floor = s.add_mesh(S.make_cube(4))
    bunny = s.add_mesh(S.make_blob(76, 2.8, 0))
    s.add_object(floor, S.translate(S.scale(S.identity(), (8.0, 0.5, 8.0)), (0.0, -3.0, 0.0)))
    s.add_object(bunny, S.translate(S.identity(), (0.0, 2.0, 0.0)))
    if with_glass:
        t = np.zeros(1, S.TRIANGLE)
        t["v0"], t["v1"], t["v2"] = (0, 0, 0), (0.01, 0, 0), (0, 0.01, 0)
        t["materialIndex"] = 3
        g = s.add_mesh(t)
        s.add_object(g, S.translate(S.identity(), (0.0, 50.0, 60.0)))
    return s.build()


for name, sc in (("opaque", scene(False)), ("one_hidden_glass_triangle", scene(True))):
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP))
    r.render(); r.sync(); r.render_history_ms()
    for _ in range(3):
        r.render()
    r.sync()
    ms = sorted(r.render_history_ms())
    print(json.dumps({"scene": name, "kernel": r.last_kernel_name(), "kernel_ms": round(ms[1], 3)}), flush=True)
    r.close()
