"""Two C2 frames with the default kernel (for rocprofv3 passes that should not pay for bench.py's extras)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

W, H, SPP, B = 1920, 1080, 64, 4
sc = S.bunny_scene(n=76, aspect=W / H)
r = Renderer(0)
r.upload_scene(sc)
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP))
for _ in range(2):
    r.render()
r.sync()
print([round(x, 2) for x in r.render_history_ms()])
r.close()
