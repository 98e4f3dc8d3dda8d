"""rz_update_transforms alone (inverse + world boxes + TLAS rebuild on the device; the call synchronises): ms per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer

for count in (16, 64, 256, 1024, 4096):
    side = int(round(count ** 0.5))
    sc = S.instanced_scene(n=4, count=count, aspect=16 / 9)
    r = Renderer(0)
    r.upload_scene(sc)
    floor = sc.arrays[S.BIND_INSTANCES]["transform"][0].copy()
    xfs = [np.stack([floor] + S.instanced_transforms(f + 1, count, spacing=24.0 / side, obj_scale=3.0 / side)) for f in range(4)]
    r.update_transforms(xfs[0])
    ts = []
    for f in range(12):
        t = time.perf_counter()
        r.update_transforms(xfs[f % 4])
        ts.append(time.perf_counter() - t)
    print(f"{count + 1} instances: rz_update_transforms {min(ts) * 1e3:.3f} ms (median {sorted(ts)[6] * 1e3:.3f})", flush=True)
    r.close()
