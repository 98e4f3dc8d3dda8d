"""Wave scratch of a launch (rz_debug_last_plan().scratch_mib) for named workloads: scratch_mib.py [c2 c3 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

for name in (sys.argv[1:] or ["c2", "c3", "c4", "c5full", "c2g", "glassbunny", "ref64"]):
    sc, W, H, spp, b = S.named_config(name)
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    r.render()
    r.sync()
    p = r.debug_last_plan()
    print(name, {k: p[k] for k in ("scratch_mib", "grid", "per_claim", "claim_units") if k in p}, r.last_kernel_name(), flush=True)
    r.close()
