# round 5: the pop-order re-convergence prototype (RZ_ORDERED_WALK) against the C++ walk and the product (asm walk), same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_ord.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_ord.log 2>&1; echo "ord gpu tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/gputests_ord.log | tee -a $O/summary.txt
for i in 1 2; do
  for v in new cpp ord; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 2>&1 | tail -1 | tee -a $O/ab.log
  done
done
RAYZEN_HIP_SO=$L/librayzen_hip_cppprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2 > $O/prof_cpp_c2.log 2>&1
RAYZEN_HIP_SO=$L/librayzen_hip_ordprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2 > $O/prof_ord_c2.log 2>&1
RAYZEN_HIP_SO=$L/librayzen_hip_cppprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2close > $O/prof_cpp_c2close.log 2>&1
RAYZEN_HIP_SO=$L/librayzen_hip_ordprof.so timeout -k 10 300 python profiles/scripts/prof_frame.py c2close > $O/prof_ord_c2close.log 2>&1
echo done
