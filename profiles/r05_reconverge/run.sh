# round 5: the pop-order re-convergence prototype, second attempt (wave-min by groups, no lane shuffles); every step gated on the one before
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
L=$PWD/rayzen_amd/lib
export RAYZEN_HIP_SO=$L/librayzen_hip_ord.so
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_gpu_cases.py -m gpu -x -q > $O/gputests_ord_small.log 2>&1 || { echo "ord small tests FAILED"; tail -5 $O/gputests_ord_small.log; exit 1; }
tail -1 $O/gputests_ord_small.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_ord.log 2>&1 || { echo "ord full tests FAILED"; tail -5 $O/gputests_ord.log; exit 1; }
tail -1 $O/gputests_ord.log
for i in 1 2; do
  for v in new cpp ord; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
RAYZEN_HIP_SO=$L/librayzen_hip_ordprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2 > $O/prof_ord_c2.log 2>&1 || exit 1
RAYZEN_HIP_SO=$L/librayzen_hip_cppprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2close > $O/prof_cpp_c2close.log 2>&1 || exit 1
RAYZEN_HIP_SO=$L/librayzen_hip_ordprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2close > $O/prof_ord_c2close.log 2>&1 || exit 1
echo done
