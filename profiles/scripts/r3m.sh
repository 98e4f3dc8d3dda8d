cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3m; mkdir -p $O
for w in c2 c4; do
python3 profiles/scripts/pmc_collect.py $O/pmc_late_$w.json rz_late_generation -- python3 profiles/scripts/one_frame.py $w > $O/pmc_late_$w.log 2>&1
done
ls $O
