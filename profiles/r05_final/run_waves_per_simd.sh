# round 5, second session: 5 and 6 waves per SIMD for the render kernels (w5: 96 VGPRs, 91 spilled; w6: 80 VGPRs, 131 spilled) with the
# persistent grid and the LDS window sized to match, against the product (4 waves, 128 VGPRs, 33 spilled)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5aj; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_w5.so timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_golden.py -m gpu -x -q > $O/gputests_w5.log 2>&1; echo "w5 gpu tests rc=$?"; tail -n 1 $O/gputests_w5.log
for i in 1 2 3; do
  for v in new w5 w6; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
