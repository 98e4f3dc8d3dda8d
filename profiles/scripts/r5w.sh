# round 5, second session: parking in front of the SECOND segment (pb1: RZ_PARK_BOUNCE=1) against the product (third segment)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5w; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_pb1.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_gpu_cases.py -m gpu -x -q > $O/gputests_pb1.log 2>&1; echo "pb1 gpu tests rc=$?"; tail -n 2 $O/gputests_pb1.log
for i in 1 2 3; do
  for v in new pb1; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref16 ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
# ... and the N > 1 bench path rehearsed on the one GPU with the final tree (all members on device 0: not a scaling number)
unset RAYZEN_HIP_SO
timeout -k 10 300 python bench.py --gpus 2 --loopback --steps 3 --warmup 1 > $O/bench_n2_loopback.json 2> $O/bench_n2_loopback.err; echo "n2 loopback rc=$?"; tail -c 700 $O/bench_n2_loopback.json
