"""Which order may a process load torch (which bundles its own libamdhip64) and librayzen_hip.so (linked against /opt/rocm's)?
hip_runtime_order.py torch-first | lib-first"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
order = sys.argv[1]


def use_lib():
    from rayzen_amd import _lib
    n = _lib.hip().rz_device_count()
    print("librayzen_hip: devices", n, flush=True)
    import numpy as np
    from rayzen_amd import scene as S
    from helpers import hip_render
    img = hip_render(S.cornell_scene(), 32, 32, 2, 2)
    print("librayzen_hip: rendered, sum", float(np.asarray(img).sum()), flush=True)


def use_torch():
    import torch
    print("torch: available", torch.cuda.is_available(), "count", torch.cuda.device_count(), flush=True)
    torch.cuda.set_device(0)
    x = torch.ones(4, device="cuda") * 2
    print("torch: tensor on device, sum", float(x.sum()), flush=True)


(use_torch, use_lib)[order == "lib-first"]()
(use_lib, use_torch)[order == "lib-first"]()
print("maps:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l or "librccl" in l}), flush=True)
