# round 5: the pool's B phase in pair mode (two lanes per walk) against the one-lane-per-walk form, same sources otherwise
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_pairs.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_gpu_cases.py tests/test_fuzz_gpu.py tests/test_gpu_configs_full.py -m gpu -x -q > $O/tests_pairs.log 2>&1 || { echo "pairs tests FAILED"; tail -15 $O/tests_pairs.log; exit 1; }
echo "pairs: $(tail -1 $O/tests_pairs.log)"
for i in 1 2 3; do
  for v in nopairs pairs; do
    export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 ref16 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
echo done
