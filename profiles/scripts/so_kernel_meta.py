#!/usr/bin/env python3
"""Register / spill / scratch figures of the render kernels as recorded in the code objects of a BUILT librayzen_hip*.so
(the AMDGPU metadata note: what the loader sees), plus static counts of spill traffic in the disassembly.

    python3 profiles/scripts/so_kernel_meta.py rayzen_amd/lib/librayzen_hip.so [substring of the mangled kernel name ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1]
want = sys.argv[2:] or ["rz_render_samplesILb0ELb0ELb0ELi8E", "rz_render_samplesILb0ELb1ELb0ELi8E", "rz_render_samplesILb0ELb0ELb0ELi16E", "rz_render_samplesILb0ELb1ELb0ELi0E"]
with tempfile.TemporaryDirectory() as tmp:
    c = shutil.copy(so, os.path.join(tmp, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", c], cwd=tmp, check=True, capture_output=True)
    cos = sorted((p for p in os.listdir(tmp) if "gfx950" in p), key=lambda p: -os.path.getsize(os.path.join(tmp, p)))
    co = os.path.join(tmp, cos[0])
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
kern = {}
for blk in notes.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
    kern[g("name")] = {k: g(k) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size")}
funcs = {}
cur = None
for l in dis.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur and l.startswith("\t"):
        funcs[cur].append(l.split()[0])
for name, meta in kern.items():
    if not any(w in name for w in want):
        continue
    ins = funcs.get(name, [])
    c = lambda op: sum(1 for i in ins if i.startswith(op))
    print(f"{os.path.basename(so)} {name[:60]:60s} VGPR {meta['vgpr_count']} spilled {meta['vgpr_spill_count']} | SGPR spilled {meta['sgpr_spill_count']} | scratch {meta['private_segment_fixed_size']} B | "
          f"insts {len(ins)} v_readlane {c('v_readlane')} v_writelane {c('v_writelane')} s_nop {c('s_nop')} scratch_ld/st {c('scratch_load')}/{c('scratch_store')} s_mov {c('s_mov_b32')}")
