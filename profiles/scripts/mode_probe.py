"""Is the 17.6 / 18.6 ms bimodality per process or per allocation?  Several contexts in ONE process, each with its own
scene buffers (the earlier contexts stay alive so that addresses differ), timed back to back."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

W, H, SPP, B = 1920, 1080, 64, 4
sc = S.bunny_scene(n=76, aspect=W / H)
keep = []
for i in range(5):
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP))
    r.render(); r.sync(); r.render_history_ms()
    for _ in range(4):
        r.render()
    r.sync()
    ms = r.render_history_ms()
    print(i, [round(x, 2) for x in ms], flush=True)
    keep.append(r)
