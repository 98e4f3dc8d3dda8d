import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params
def make(mat):
    s = S.Scene(camera=S.Camera(position=(0.0, 2.5, 10.0), aspect=16/9))
    floor = s.add_mesh(S.make_cube(4)); bunny = s.add_mesh(S.make_blob(76, 2.8, 0))
    s.add_object(floor, S.translate(S.scale(S.identity(), (8.0, 0.5, 8.0)), (0.0, -3.0, 0.0)))
    s.add_object(bunny, S.translate(S.identity(), (0.0, 2.0, 0.0)))
    t = np.zeros(1, S.TRIANGLE); t["v0"], t["v1"], t["v2"] = (0, 0, 0), (0.01, 0, 0), (0, 0.01, 0); t["materialIndex"] = mat
    s.add_object(s.add_mesh(t), S.translate(S.identity(), (0.0, 50.0, 60.0)))
    return s.build()
for name, mat in (("hidden triangle OPAQUE (opaque kernels)", 0), ("hidden triangle GLASS (transparent kernels)", 3)):
    sc = make(mat)
    r = Renderer(0); r.upload_scene(sc)
    r.set_frame(frame_params(sc.camera, 1920, 1080, len(sc.lights), 4, 64))
    for _ in range(4): r.render()
    r.sync(); ms = r.render_history_ms()[1:]
    print(name, f"{min(ms):.3f} ms", r.last_kernel_name(), flush=True); r.close()
