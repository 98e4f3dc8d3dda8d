cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ad; mkdir -p $O
RZ_GLASS_TWO_PASS=0 timeout -k 10 600 python3 profiles/scripts/glass_work.py > $O/work.log 2>&1 || { cat $O/work.log; exit 1; }
cat $O/work.log
