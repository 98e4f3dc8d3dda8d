set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs_full.py tests/test_gpu_cases.py tests/test_tlas_order_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2 c4 c2g ref c5 c5full"
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run A=1
CFG="c4 c5"
run RZ_SMALL_SPP_CLAIMS=0
run RZ_GROUPS_PER_CLAIM=8
run RZ_GROUPS_PER_CLAIM=16
run RZ_GROUPS_PER_CLAIM=2
CFG="c2 c4 c2g ref c5 c5full"
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run A=1
cat $O/ab.log
