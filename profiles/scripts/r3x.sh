cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3z; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { grep -v "dist-packages" $O/gpu_tests.log | tail -40; exit 1; }
tail -3 $O/gpu_tests.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2 c4 c3 c5 c5full"
run A=1
run A=1
cat $O/ab.log
