"""Device BLAS build (rz_build_blas) vs the host builder (rzh_build_blas) on the blob meshes: wall and device time."""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer

r = Renderer(0)
out = []
for n in (12, 76, 150, 289):
    t = S.make_blob(n, 1.0, 0)
    r.build_blas(t[:64])                       # warm the allocator / code objects
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        nodes, idx, depth, ms = r.build_blas(t)
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, ms)
    t0 = time.perf_counter()
    hn, hi, hd = S.build_blas(t)
    host = (time.perf_counter() - t0) * 1e3
    same = nodes.tobytes() == hn.tobytes() and idx.tobytes() == hi.tobytes()
    rec = {"triangles": int(t.shape[0]), "nodes": int(nodes.shape[0]), "depth": depth, "device_ms": round(best[1], 3),
           "device_wall_ms": round(best[0], 3), "host_ms": round(host, 2), "identical": bool(same)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
r.close()
