"""Copies what collect_round5.sh gathered (gpurun_out/prof_r05/) into the tracked profiles/r05_* directories and prints the
figures the READMEs and DESIGN.md quote (run from the repository root, after the gpurun call has merged its output).  A second
run after collect_round5b.sh also installs the bench line and the bench_configs table."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r05")
INST = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")
for src, dst in (("c2", "c2_kernel"), ("c4", "c4"), ("c5full", "c5"), ("c2g", "c2g"), ("glassbunny", "glassbunny"), ("c2close", "c2close"), ("ref", "ref"), ("ref16", "ref16"), ("c3", "c3"), ("ref64", "ref64")):
    if not os.path.exists(os.path.join(SRC, f"pmc_{src}.json")):
        continue
    d = os.path.join(ROOT, "profiles", "r05_" + dst)
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(SRC, f"kt_{src}", "run_kernel_stats.csv"), os.path.join(d, "kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, f"pmc_{src}.json"), os.path.join(d, "pmc_rz_render_samples.json"))
    if os.path.exists(os.path.join(SRC, f"counters_{src}.json")):      # the launch's algorithmic tallies (tests/test_workmodel.py prices them)
        shutil.copy(os.path.join(SRC, f"counters_{src}.json"), os.path.join(d, "counters.json"))
    p = json.load(open(os.path.join(d, "pmc_rz_render_samples.json")))
    ins = sum(p[k] for k in INST)
    t = p["_dispatch"]["duration_ns_under_profiler"] * 1e-9
    rows = [r for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))) if "rz_" in r["Name"] and "render" in r["Name"]]
    live = p["SQ_INSTS_VALU"] * p["SQ_THREAD_CYCLES_VALU"] / p["SQ_ACTIVE_INST_VALU"]
    print(dst, "hash", p["_source_hash"][:16], "|", "; ".join("%s x%s avg %.3f ms" % (r["Name"].split("(")[0].replace("void rz::", ""), r["Calls"], float(r["AverageNs"]) / 1e6) for r in rows),
          "| inst %.4g" % ins, " ".join("%s %.3g" % (k[9:].lower(), p[k]) for k in INST), "| ms(pmc) %.3f" % (t * 1e3),
          "issue frac %.3f" % (ins / t / 1e9 / 1228.8),
          "| lane util %.3f" % (p["SQ_THREAD_CYCLES_VALU"] / (64 * p["SQ_ACTIVE_INST_VALU"])),
          "| useful %.3f" % (live / t / (1024 * 32 * 2.4e9)),
          "| HBM GB %.2f (%.0f GB/s)" % ((2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024 / 1e9, (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024 / 1e9 / t),
          "| tcc hit %.3f" % (p["TCC_HIT_sum"] / (p["TCC_HIT_sum"] + p["TCC_MISS_sum"])),
          "| TA busy %.2f" % (p["TA_TA_BUSY_sum"] / (256 * t * 2.4e9)),
          "| icache req %.4g miss %.4g hit %.5f" % (p.get("SQC_ICACHE_REQ", 0), p.get("SQC_ICACHE_MISSES", 0), p.get("SQC_ICACHE_HITS", 0) / max(p.get("SQC_ICACHE_REQ", 1), 1)),
          "| scratch %s" % p["_dispatch"]["Scratch_Size"])
for f, dst in (("bench_c2.json", ("r05_c2_kernel", "bench.json")), ("bench_configs.log", ("r05_c4", "bench_configs.log")), ("rank_share.log", ("r05_c2_kernel", "rank_share.log"))):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(ROOT, "profiles", dst[0], dst[1]))
