"""The shader's own work (rz_render_counted's tallies, priced by bench.py's table) of the transparent frames beside C2's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params

cfgs = {"c2": lambda: (S.bunny_scene(n=76, aspect=16 / 9), 1920, 1080, 64, 4),
        "c2g": lambda: (S.bunny_scene(n=76, aspect=16 / 9, extras=True), 1920, 1080, 64, 4),
        "glassbunny": lambda: (S.bunny_scene(n=76, aspect=16 / 9, bunny_material=3), 1920, 1080, 64, 4)}
for name in (sys.argv[1:] or ["c2", "c2g", "glassbunny"]):
    sc, W, H, spp, b = cfgs[name]()
    r = Renderer(0)
    r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, W, H, len(sc.lights), b, spp))
    cnt = r.render_counted()
    for _ in range(3):
        r.render()
    r.sync()
    ms = min(r.render_history_ms()[-2:])
    wm = bench.work_model(cnt, len(sc.lights), ms * 1e-3)
    print(name, f"{ms:.3f} ms", {k: cnt[k] for k in cnt}, "lane_ops %.3e" % wm["lane_ops"], "frac", wm["frac"], flush=True)
    r.close()
