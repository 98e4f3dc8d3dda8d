"""Minimum kernel ms of named workloads (rayzen_amd/scene.py: NAMED_CONFIGS) for the loaded library (RAYZEN_HIP_SO selects a
variant): the A/B workhorse.  config_ms.py [c2 c4 c5 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

out = []
for name in (sys.argv[1:] or ["c2", "c4", "c5"]):
    sc, W, H, spp, b = S.named_config(name)
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    for _ in range(4):
        r.render()
    r.sync()
    ms = r.render_history_ms()[1:]
    out.append(f"{name} {min(ms):.3f}")
    r.close()
print(os.path.basename(os.environ.get("RAYZEN_HIP_SO", "new")), " ".join(out), flush=True)
