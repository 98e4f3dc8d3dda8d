import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer
r = Renderer(0)
rng = np.random.default_rng(7)
bad = 0
for k in range(400):
    n = int(rng.choice([rng.integers(1, 70), rng.integers(60, 5000), rng.integers(4000, 60000)]))
    t = np.zeros(n, S.TRIANGLE)
    mode = k % 4
    c = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    if mode == 1: c = np.round(c)                       # many equal centroids
    if mode == 2: c[:, 1] = 0.0                         # planar
    for key in ("v0", "v1", "v2"):
        t[key] = c + (rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32) if mode != 3 else np.round(rng.uniform(-1, 1, (n, 3))).astype(np.float32))
    nodes, idx, depth, ms = r.build_blas(t)
    hn, hi, hd = S.build_blas(t)
    if nodes.tobytes() != hn.tobytes() or idx.tobytes() != hi.tobytes() or depth != hd:
        bad += 1; print("MISMATCH", k, n, mode)
print("done, mismatches:", bad)
