# round 5: the two hash numbers of a camera path (and of a hemisphere draw) evaluated side by side, their binary64 constants materialised once
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_rand2.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_gpu_cases.py tests/test_fuzz_gpu.py tests/test_gpu_configs_full.py -m gpu -x -q > $O/tests_rand2.log 2>&1 || { echo "rand2 tests FAILED"; tail -15 $O/tests_rand2.log; exit 1; }
echo "rand2: $(tail -1 $O/tests_rand2.log)"
for i in 1 2 3; do
  for v in new rand2; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 ref16 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
