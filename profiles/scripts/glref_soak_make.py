"""One-off soak, part 1 (build container): N seeded random scenes rendered by RayZen's own shader (oracle/glref, Mesa llvmpipe) into
gpurun_in/glref_soak/*.npz (git-ignored; the directory travels to the GPU box with the snapshot).  Part 2, on the GPU:
glref_soak_gpu.py renders the same inputs with the HIP path and compares.  usage: glref_soak_make.py [first_seed [count]]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rayzen_amd import scene as S          # noqa: E402
from oracle.glref import glref             # noqa: E402
from test_fuzz_gpu import random_scene     # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
out = os.path.join(ROOT, "gpurun_in", "glref_soak")
os.makedirs(out, exist_ok=True)
for seed in range(first, first + count):
    sc, rng = random_scene(seed)
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 65))
    spp, b = int(rng.choice([1, 2, 3, 8])), int(rng.integers(1, 9))
    sc.camera.aspect = W / H
    sc.camera.update()
    img, info = glref.render_scene(sc, W, H, b, num_samples=spp)
    cam = sc.camera
    data = {f"b{k}": np.frombuffer(np.ascontiguousarray(sc.arrays[k]).tobytes(), np.uint8) for k in S.BINDING_DTYPES}
    data.update(cam_view=cam.view, cam_proj=cam.proj, cam_inv_view=cam.inv_view, cam_inv_proj=cam.inv_proj, cam_pos=np.asarray(cam.position, np.float32),
                out0=np.ascontiguousarray(img[..., :3]), renders=np.array(json.dumps([dict(W=W, H=H, budget=b, spp=spp)])), gl=np.array(info))
    np.savez_compressed(os.path.join(out, f"glref_soak{seed}.npz"), **data)
print("made", count, "scenes in", out)
