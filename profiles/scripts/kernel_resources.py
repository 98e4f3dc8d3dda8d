#!/usr/bin/env python3
"""Register / scratch / occupancy of every instantiation of the render kernels, as hipcc reports them
(-Rpass-analysis=kernel-resource-usage), for the sources in the tree (or the directory given first) + extra hipcc flags.

    python3 profiles/scripts/kernel_resources.py [SRC_DIR] [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
H = args.pop(0) if args and os.path.isdir(args[0]) else os.path.join(ROOT, "rayzen_amd", "csrc", "hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fno-slp-vectorize", *args, "-I", os.path.join(ROOT, "include"), "-I", H, "-c", os.path.join(H, "rz_kernels.hip"),
       "-o", "/tmp/rz_kernels_res.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void rz::|\(rz::KParams.*", "", name)
    g = lambda *names: next((v[n] for n in names if n in v), -1)
    print(f"{name:46s} VGPRs {g('VGPRs'):4d} spilled {g('VGPRs Spill', 'VGPR Spill'):3d}  SGPRs {g('TotalSGPRs', 'SGPRs'):4d} spilled {g('SGPRs Spill'):3d}  "
          f"scratch B/lane {g('ScratchSize [bytes/lane]'):5d}  waves/SIMD {g('Occupancy [waves/SIMD]'):2d}")
