# round 5, second session: sin_ (one result) and sincos_pair (two results) as out-of-line functions (sn2) against the shared five-result
# core behind one call (sn0 = the tree's default, sources refactored); parity of sn2 on three test files first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5ag; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_sn2.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_glref.py tests/test_golden.py tests/test_gpu_cases.py -m gpu -x -q > $O/gputests_sn2.log 2>&1; echo "sn2 gpu tests rc=$?"; tail -n 2 $O/gputests_sn2.log
for i in 1 2 3; do
  for v in sn0 sn2; do
    export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref16 ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
