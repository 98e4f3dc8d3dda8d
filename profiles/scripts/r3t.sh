set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3t; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs_full.py -x -q -m gpu -k "forking" > $O/gpu_tests.log 2>&1 || { grep -v "dist-packages" $O/gpu_tests.log | tail -40; exit 1; }
tail -3 $O/gpu_tests.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2g glassbunny"
run A=1
run RZ_GLASS_FORK=0
run RZ_WPOOL_CHUNK=512
run RZ_WPOOL_CHUNK=128
cat $O/ab.log
