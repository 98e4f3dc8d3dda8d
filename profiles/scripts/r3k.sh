cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3k; mkdir -p $O
for v in rf2 rf24 rf48; do
for w in c2 c4; do
RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v}_$w -o run -- python3 profiles/scripts/one_frame.py $w > $O/kt_${v}_$w.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt_${v}_$w/run_kernel_stats.csv")))
for r in rows[:2]: print("$v $w", r["Name"][:50], r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3))
PY
done; done
