#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters for one kernel, one counter group per pass, and write a JSON summary.

    python3 profiles/scripts/pmc_collect.py OUT.json KERNEL_SUBSTRING [--groups a,b,c ...] -- python3 target.py ...

Every group runs in its own `rocprofv3 --pmc ... -- <cmd>` pass (SQ has 8 slots per pass, TCC 4 with FETCH_SIZE
costing 3 and WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with a trace domain.  The
summary keeps, per counter, the value of the LAST dispatch whose kernel name contains KERNEL_SUBSTRING (per-dispatch
values, summed over the chip's XCDs by rocprofv3) plus the dispatch's grid / workgroup / VGPR / scratch fields.
rocprofv3 itself is started as a child process (the program after `--` is run directly by rocprofv3, no shell hop)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

DEFAULT_GROUPS = [
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH",
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU",
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE",
    "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64",
    "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VSKIPPED SQ_LDS_BANK_CONFLICT",
    "FETCH_SIZE TCC_HIT_sum",
    "WRITE_SIZE TCC_MISS_sum",
    "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum",
    # the instruction cache (a 99-KB kernel whose waves sit in different phases); one pass of their own, so
    # that a build of rocprofv3 that lacks one of the names loses only this pass
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE",
    "SQ_IFETCH SQ_IFETCH_LEVEL",
]


def main():
    if "--" not in sys.argv:
        raise SystemExit(__doc__)
    k = sys.argv.index("--")
    args, cmd = sys.argv[1:k], sys.argv[k + 1:]
    out_json, kernel = args[0], args[1]
    groups = DEFAULT_GROUPS
    workload = None
    if "--workload" in args:            # e.g. --workload 1920,1080,64,4,76  (recorded as _workload; bench.py matches on it)
        i = args.index("--workload")
        workload = [int(x) for x in args[i + 1].split(",")]
        del args[i:i + 2]
    if "--groups" in args:
        groups = [g.replace(",", " ") for g in args[args.index("--groups") + 1:]]
    tmp_root = os.path.join(os.path.dirname(os.path.abspath(out_json)) or ".", "_pmc_tmp")
    summary, meta = {}, {}
    for gi, g in enumerate(groups):
        d = os.path.join(tmp_root, f"g{gi}")
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        log = open(os.path.join(d, "log.txt"), "w")
        rc = subprocess.call(["rocprofv3", "--pmc", *g.split(), "--output-format", "csv", "-d", d, "-o", "run", "--"] + cmd,
                             stdout=log, stderr=subprocess.STDOUT)
        log.close()
        print(f"[pmc] pass {gi} ({g}): exit {rc}", flush=True)
        if rc != 0:
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            last = {}
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"]:
                    last.setdefault(row["Dispatch_Id"], []).append(row)
            if not last:
                continue
            did = max(last, key=lambda x: int(x))
            for row in last[did]:
                summary[row["Counter_Name"]] = summary.get(row["Counter_Name"], 0.0) * 0 + float(row["Counter_Value"])
                meta = {"Kernel_Name": row["Kernel_Name"], "Grid_Size": int(row["Grid_Size"]), "Workgroup_Size": int(row["Workgroup_Size"]),
                        "LDS_Block_Size": int(row["LDS_Block_Size"]), "Scratch_Size": int(row["Scratch_Size"]),
                        "VGPR_Count": int(row["VGPR_Count"]), "SGPR_Count": int(row["SGPR_Count"]),
                        "duration_ns_under_profiler": int(row["End_Timestamp"]) - int(row["Start_Timestamp"])}
    summary["_dispatch"] = meta
    summary["_command"] = " ".join(cmd)
    summary["_workload"] = workload
    try:                                 # which build the counters belong to (rayzen_amd/build.py: source_hash)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from rayzen_amd import _lib as rzlib        # the hash compiled into the library the profiled command loads (RAYZEN_HIP_SO is honoured)
        summary["_source_hash"] = rzlib.hip().rz_source_hash().decode()
        summary["_library"] = os.path.relpath(rzlib.HIP_SO)
    except Exception as e:
        summary["_source_hash"] = f"unavailable: {e}"
    json.dump(summary, open(out_json, "w"), indent=1)
    print(f"[pmc] wrote {out_json}: {sum(1 for k in summary if not k.startswith('_'))} counters", flush=True)
    shutil.rmtree(tmp_root, ignore_errors=True)


if __name__ == "__main__":
    main()
