"""Two frames of one named workload with the default kernel (for rocprofv3 passes that should not pay for bench.py's
extras):  one_frame.py [c2|c2close|c2g|glassbunny|c3|c4|c5|c5full|ref|...]   (rayzen_amd/scene.py: NAMED_CONFIGS; c2 is the bench
workload, c5 = C5 at 32 spp, c5full at its stated 128 spp, ref = RayZen's own scene at 800x600, 1 spp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
COUNTERS = sys.argv[sys.argv.index("--counters") + 1] if "--counters" in sys.argv else None      # also write rz_render_counted's tallies here (JSON)
if which not in S.NAMED_CONFIGS:
    raise SystemExit(which)
sc, W, H, SPP, B = S.named_config(which)
r = Renderer(0)
r.upload_scene(sc)
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, SPP))
for _ in range(2):
    r.render()
r.sync()
print(which, [round(x, 3) for x in r.render_history_ms()])
if COUNTERS:
    import json
    ms = r.render_history_ms()
    cnt = r.render_counted()
    json.dump({"config": which, "width": W, "height": H, "spp": SPP, "bounces": B, "counters": cnt}, open(COUNTERS, "w"))
r.close()
