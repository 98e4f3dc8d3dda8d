#!/usr/bin/env python3
"""Where does the one-lane-per-sample kernel lose lanes, and what would cross-pixel compaction of the late bounces buy?

    python tests/analysis/late_bounce_model.py [c2|c4|c5]         (CPU only, ~1 minute; uses the oracle's trace recorder)

Model: a wavefront = the 64 samples of one pixel; it runs its paths' closest-hit queries in lock step, query k of every
path together (what rz_render_samples does), and a query round costs the wave the MAXIMUM of its lanes' query costs
(cost of a query = BLAS nodes/2 + 2 x triangles + 3, i.e. roughly its wave-level instruction count -- the kernel is
bound by instruction issue).  Printed per round: the lanes' summed work, the lock-step cost, and
the cost if the live lanes of the 8 pixels of one persistent claim were first compacted into full waves.
Not a pytest file (no test_ prefix): an analysis tool whose output HISTORY.md section 7 quotes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rzo                                              # noqa: E402
from rayzen_amd import scene as S                                   # noqa: E402
from helpers import oracle_frame, oracle_scene                      # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H = 480, 270                                                     # a quarter of 1080p in each direction: same statistics
if which == "c2":
    sc, spp, b = S.bunny_scene(n=76, aspect=16 / 9), 64, 4
elif which == "c4":
    sc, spp, b = S.instanced_scene(n=76, count=16, aspect=16 / 9), 64, 4      # 64 spp: one pixel per wave, as in C2
else:
    sc, spp, b = S.stress_scene(n=289, aspect=16 / 9), 64, 8
rzo.build()
L = rzo.lib()
L.rzo_set_trace_recorder.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
rec = np.zeros((H * W, spp, 8), np.uint16)
L.rzo_set_trace_recorder(rec.ctypes.data, W, H, spp)
rzo.render(oracle_scene(sc), oracle_frame(sc, W, H, spp, b), nthreads=os.cpu_count() or 8)
L.rzo_set_trace_recorder(None, 0, 0, 0)
rec = rec.astype(np.int64)
work, lock = rec.sum(), 0
print(f"{which}: {W}x{H}, {spp} spp, {b} bounces; query rounds 0 = primary, then shadow queries, then bounces")
print("round   lanes' work   lock-step cost   utilisation   live lanes/wave   + 8-pixel compaction")
tot_compact = 0
for k in range(8):
    c = rec[:, :, k]
    mx = c.max(axis=1)
    if mx.sum() == 0:
        continue
    cost = int(mx.sum()) * 64
    lock += cost
    live = (c > 0).sum(axis=1)
    groups = c.reshape(H, W // 8, 8 * spp).reshape(-1, 8 * spp)     # the 8 consecutive pixels of one claim
    comp = 0
    for g in groups:
        l = g[g > 0]
        for i in range(0, len(l), 64):
            comp += int(l[i:i + 64].max()) * 64
    tot_compact += min(comp, cost)
    print(f"{k:5d} {c.sum():13d} {cost:16d} {c.sum() / cost:13.3f} {live[mx > 0].mean():17.1f} {comp:22d}")
print(f"total  {work:13d} {lock:16d} {work / lock:13.3f}                     {tot_compact:22d}  ({(1 - tot_compact / lock) * 100:.1f} % fewer wave-steps)")
