"""A transparent frame rendered by the speculating kernel alone and in two passes (opaque kernel + speculating kernel over the
marked groups): kernel ms of both, the share of marked groups, and whether the two frames hold the same bits."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

cfgs = {"c2g": lambda: (S.bunny_scene(n=76, aspect=16 / 9, extras=True), 1920, 1080, 64, 4),
        "glassbunny": lambda: (S.bunny_scene(n=76, aspect=16 / 9, bunny_material=3), 1920, 1080, 64, 4),
        "c2g16": lambda: (S.bunny_scene(n=76, aspect=16 / 9, extras=True), 1920, 1080, 16, 4),
        "c2g256": lambda: (S.bunny_scene(n=76, aspect=16 / 9, extras=True), 1920, 1080, 256, 4)}
for name in (sys.argv[1:] or ["c2g", "glassbunny"]):
    sc, W, H, spp, b = cfgs[name]()
    frames = {}
    line = [name]
    for mode in ("0", "old", "1"):
        os.environ["RZ_GLASS_TWO_PASS"] = "0" if mode == "0" else "1"
        os.environ["RZ_GLASS_RESOLVE"] = "0" if mode == "old" else "1"
        r = Renderer(0)
        r.upload_scene(sc)
        r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
        for _ in range(4):
            r.render()
        r.sync()
        ms = r.render_history_ms()[1:]
        frames[mode] = r.read_accum().copy()
        line.append(f"{ {'0': 'single', 'old': 'two-pass, group code', '1': 'two-pass, resolve'}[mode]} {min(ms):.3f} ms [{r.last_kernel_name()}]")
        r.close()
    same = np.array_equal(frames["0"].view(np.uint32), frames["1"].view(np.uint32)) and np.array_equal(frames["0"].view(np.uint32), frames["old"].view(np.uint32))
    line.append(f"same bits {same}")
    if not same:
        d = np.any(frames["0"].view(np.uint32) != frames["1"].view(np.uint32), axis=-1)
        line.append(f"{int(d.sum())} pixels differ")
    print(" | ".join(line), flush=True)
