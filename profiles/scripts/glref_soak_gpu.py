"""One-off soak, part 2 (GPU box): every gpurun_in/glref_soak/*.npz (inputs + the frame RayZen's own shader rendered for them on Mesa
llvmpipe, made by glref_soak_make.py) through the HIP path behind the C-ABI; rz_present's float output against the shader's FragColor,
and against the oracle bit for bit."""
import glob
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rayzen_amd import scene as S                          # noqa: E402
from rayzen_amd.renderer import Renderer, frame_params     # noqa: E402
from oracle import rzo                                     # noqa: E402
from helpers import oracle_scene, oracle_frame, sync_oracle_flavour   # noqa: E402

flavour = sync_oracle_flavour()
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_in", "glref_soak", "*.npz")))
tot = far = bit = worst_scene = same_as_oracle = 0
by_budget = {}
R = Renderer(0)
for f in files:
    z = np.load(f)
    arrays = {b: np.frombuffer(z[f"b{b}"].tobytes(), dt).copy() for b, dt in S.BINDING_DTYPES.items()}
    cam = types.SimpleNamespace(view=z["cam_view"], proj=z["cam_proj"], inv_view=z["cam_inv_view"], inv_proj=z["cam_inv_proj"], position=z["cam_pos"])
    sc = types.SimpleNamespace(arrays=arrays, camera=cam, lights=arrays[S.BIND_LIGHTS])
    r = json.loads(str(z["renders"]))[0]
    want = z["out0"]
    R.upload_scene(sc)
    R.set_frame(frame_params(cam, r["W"], r["H"], len(sc.lights), r["budget"], r["spp"]))
    R.render()
    R.sync()
    rgb, _ = R.present()
    osc = oracle_scene(sc)
    acc = rzo.render(osc, oracle_frame(sc, r["W"], r["H"], r["spp"], r["budget"]), nthreads=16)
    ref, _ = rzo.present(osc, acc, cam.view, cam.proj, len(sc.lights))
    same_as_oracle += int((rgb.view(np.uint32) == ref.view(np.uint32)).all())
    d = np.abs(rgb.astype(np.float64) - want.astype(np.float64)).max(axis=-1)
    n_far = int((d > 1e-4).sum())
    tot += d.size
    far += n_far
    bit += int((rgb.view(np.uint32) == want.view(np.uint32)).all(axis=-1).sum())
    worst_scene = max(worst_scene, n_far / d.size)
    bb = by_budget.setdefault(r["budget"], [0, 0])
    bb[0] += d.size
    bb[1] += n_far
    print(f"{os.path.basename(f)} {r}: Linf {d.max():.2e}  beyond 1e-4: {n_far} of {d.size}", flush=True)
R.close()
print(f"TOTAL (library math flavour {flavour}): {len(files)} random scenes, {tot} pixels; HIP vs RayZen's shader: {far} pixels beyond 1e-4 ({far / max(tot, 1) * 100:.4f} %), "
      f"{bit / max(tot, 1) * 100:.1f} % bit-identical, worst scene {worst_scene * 100:.2f} %; HIP == oracle bit for bit on {same_as_oracle} of {len(files)} scenes")
print("by bounce budget (pixels, beyond 1e-4):", {k: tuple(v) for k, v in sorted(by_budget.items())})
