# round 5, second session: advance() told what cannot happen at its two call sites in the compacting kernels (ph1: RZ_PHASES=1, the
# tree's default) against the same sources without (ph0); parity of ph1 on four test files first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5x; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_ph1.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_glref.py tests/test_golden.py tests/test_gpu_cases.py tests/test_gpu_configs_full.py -m gpu -x -q > $O/gputests_ph1.log 2>&1; echo "ph1 gpu tests rc=$?"; tail -n 2 $O/gputests_ph1.log
for i in 1 2 3; do
  for v in ph0 ph1; do
    export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref16 ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
