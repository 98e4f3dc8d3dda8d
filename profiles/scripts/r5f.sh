# round 5, batch 4: latency experiments (two triangles per wait; right-child pair touched ahead) + cold IEEE paths out of line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
L=$PWD/rayzen_amd/lib
for v in tri2 pfr slowni; do
  RAYZEN_HIP_SO=$L/librayzen_hip_$v.so timeout -k 10 400 python -m pytest tests/test_parity_gpu.py tests/test_gpu_cases.py tests/test_gpu_configs_full.py -m gpu -x -q > $O/tests_$v.log 2>&1 || { echo "$v tests FAILED"; tail -5 $O/tests_$v.log; exit 1; }
  echo "$v: $(tail -1 $O/tests_$v.log)"
done
for i in 1 2 3; do
  for v in new tri2 pfr slowni; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
echo done
