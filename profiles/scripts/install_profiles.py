"""Copies what collect_round.sh gathered (gpurun_out/prof_r02b/) into the tracked profiles/r02b_* directories and prints
the figures the READMEs and DESIGN.md quote (run from the repository root, after the gpurun call has merged its output)."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02b")
INST = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")
for src, dst in (("c2", "c2_kernel"), ("c4", "c4"), ("c5full", "c5")):
    d = os.path.join(ROOT, "profiles", "r02b_" + dst)
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(SRC, f"kt_{src}", "run_kernel_stats.csv"), os.path.join(d, "kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, f"pmc_{src}.json"), os.path.join(d, "pmc_rz_render_samples.json"))
    p = json.load(open(os.path.join(d, "pmc_rz_render_samples.json")))
    ins = sum(p[k] for k in INST)
    t = p["_dispatch"]["duration_ns_under_profiler"] * 1e-9
    first = next(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
    print(dst, "hash", p["_source_hash"][:16], "| kernel_stats %s calls avg ms %.3f min %.3f" % (first["Calls"], float(first["AverageNs"]) / 1e6, float(first["MinNs"]) / 1e6),
          "| inst %.4g" % ins, " ".join("%s %.3g" % (k[9:].lower(), p[k]) for k in INST), "| ms(pmc) %.3f" % (t * 1e3),
          "G/s %.1f frac %.3f" % (ins / t / 1e9, ins / t / 1e9 / 1228.8),
          "| lane util %.3f" % (p["SQ_THREAD_CYCLES_VALU"] / (64 * p["SQ_ACTIVE_INST_VALU"])),
          "| HBM GB %.2f (%.0f GB/s)" % ((2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024 / 1e9, (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024 / 1e9 / t),
          "| tcc hit %.3f" % (p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])))
