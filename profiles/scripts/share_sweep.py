"""rank 0's share of C2 split over N ranks (strong scaling), under the environment's launch knobs: share_sweep.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, W, H, spp, b = S.named_config("c2")
r = Renderer(0); r.upload_scene(sc)
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp, 0, 0, n))
for _ in range(5): r.render()
r.sync()
print(f"N={n} share {min(r.render_history_ms()[1:]):.3f} ms plan {r.debug_last_plan()}", flush=True)
r.close()
