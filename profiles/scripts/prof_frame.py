"""One counted C2 frame with the RZ_PROF diagnostic build (RAYZEN_HIP_SO=rayzen_amd/lib/librayzen_hip_prof.so):
prints per-site wave executions / active lanes and the wave-cycle split to stderr."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

W, H, SPP, B = 1920, 1080, 64, 4
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
NRANKS = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # render rank 0's share of an NRANKS-way tile split
if which == "c2":
    sc = S.bunny_scene(n=76, aspect=W / H)
elif which == "c4":
    sc, SPP = S.instanced_scene(n=76, count=16, aspect=W / H), 16
elif which == "c2g":
    sc = S.bunny_scene(n=76, aspect=W / H, extras=True)
elif which == "ref":        # RayZen's own workload: 800x600, 1 spp, 5 bounces (main.cpp:35-36, 600; FS:675)
    W, H, SPP, B = 800, 600, 1, 5
    sc = S.reference_scene(aspect=W / H)
elif which in S.NAMED_CONFIGS and which not in ("c2", "c4", "c2g", "ref", "c5"):
    sc, W, H, SPP, B = S.named_config(which)
elif which == "c5":
    W, H, SPP, B = 3840, 2160, 16, 8
    sc = S.stress_scene(n=289, aspect=W / H)
r = Renderer(0)
r.upload_scene(sc)
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP, 0, 0, NRANKS))
r.render()
r.sync()
c = r.render_counted()
print(which, r.render_history_ms(), c)
r.close()
