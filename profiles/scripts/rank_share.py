"""What ONE rank of an N-GPU weak-scaling run does, timed on one GPU: 1/N of the tiles at 64*N samples per pixel, as a
single launch and as N launches of 64 samples each (sample_base = 64 k; bit-identical by construction).
    python profiles/scripts/rank_share.py [N ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

W, H, B = 1920, 1080, 4
sc = S.bunny_scene(n=76, aspect=W / H)
r = Renderer(0)
r.upload_scene(sc)
for n in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    spp = 64 * n
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, spp, 0, 0, n))
    for _ in range(3):
        r.render()
    r.sync()
    one = min(r.render_history_ms()[1:])
    ref = r.read_accum().copy()
    tot = []
    for rep in range(3):
        t = 0.0
        for k in range(n):
            r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, 64, 64 * k, 0, n))
            r.render()
            r.sync()
            t += r.render_history_ms()[-1]
        tot.append(t)
    same = bool((r.read_accum().view(np.uint32) == ref.view(np.uint32)).all())
    print(f"N={n}: one launch of {spp} spp {one:.3f} ms; {n} launches of 64 spp {min(tot):.3f} ms; same bits {same}", flush=True)
# BASELINE configs[2] literally (C3): 256 spp in total, 8 ranks -- one rank's share against 1/8 of the single-GPU frame
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, 256, 0, 0, 1))
for _ in range(2):
    r.render()
r.sync()
full = min(r.render_history_ms()[1:])
r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), B, 256, 0, 0, 8))
for _ in range(3):
    r.render()
r.sync()
share = min(r.render_history_ms()[1:])
print(f"C3: whole 256-spp frame on one GPU {full:.3f} ms (1/8 = {full / 8:.3f}); rank 0 of 8: {share:.3f} ms", flush=True)
r.close()
