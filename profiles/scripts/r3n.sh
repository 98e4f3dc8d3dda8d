set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs_full.py -x -q -m gpu -k "global_pool or compacting" > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2 c4 c5 c5full c3"
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run A=1
run RZ_CROSS_CLAIM_POOL=0
CFG="c2 c4"
for ch in 128 256 1024 2048; do run RZ_WPOOL_CHUNK=$ch; done
CFG="c4"
for pc in 8 16; do run RZ_GROUPS_PER_CLAIM=$pc; done
cat $O/ab.log
