"""One device BLAS build of the 1 M-triangle blob (for rocprofv3 --kernel-trace)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 289
t = S.make_blob(n, 1.0, 0)
r = Renderer(0)
nodes, idx, depth, ms = r.build_blas(t)
print(t.shape[0], nodes.shape[0], depth, ms)
r.close()
