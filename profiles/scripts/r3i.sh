cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; mkdir -p $O
for w in c2 c4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o run -- python3 profiles/scripts/one_frame.py $w > $O/kt_$w.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt_$w/run_kernel_stats.csv")))
for r in rows[:12]: print("$w", r["Name"][:70], r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3), "tot ms %.3f"%(float(r["TotalDurationNs"])/1e6))
PY
done
