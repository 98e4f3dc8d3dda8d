"""Kernel ms of the BASELINE configurations for the loaded library (RAYZEN_HIP_SO selects a variant): c2 c4 c5 [c5 at 32 spp]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

cfgs = {"c2": lambda: (S.bunny_scene(n=76, aspect=16 / 9), 1920, 1080, 64, 4),
        "c2g": lambda: (S.bunny_scene(n=76, aspect=16 / 9, extras=True), 1920, 1080, 64, 4),
        "glassbunny": lambda: (S.bunny_scene(n=76, aspect=16 / 9, bunny_material=3), 1920, 1080, 64, 4),
        "ref": lambda: (S.reference_scene(aspect=800 / 600), 800, 600, 1, 5),
        "c4": lambda: (S.instanced_scene(n=76, count=16, aspect=16 / 9), 1920, 1080, 16, 4),
        "c5": lambda: (S.stress_scene(n=289, aspect=16 / 9), 3840, 2160, 32, 8),
        "c5full": lambda: (S.stress_scene(n=289, aspect=16 / 9), 3840, 2160, 128, 8),
        "c3": lambda: (S.bunny_scene(n=76, aspect=16 / 9), 1920, 1080, 256, 4),
        "mirror": lambda: (S.bunny_scene(n=76, aspect=16 / 9, bunny_material=2, floor_material=2), 1920, 1080, 64, 8)}
out = []
for name in (sys.argv[1:] or ["c2", "c4", "c5"]):
    sc, W, H, spp, b = cfgs[name]()
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
