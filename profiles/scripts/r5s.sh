# round 5, second session: more code out of the one big function?  shade_light's BRDF term (sh1) and the camera ray of begin_sample (bg1)
# behind calls, both (sb1); sh0 = the same sources with both inline (the refactoring alone), base = the product at HEAD.
# Parity of sb1 on a subset first (a variant must render the same bits), then the timings.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5s; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_sb1.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_glref.py tests/test_golden.py tests/test_gpu_cases.py -m gpu -x -q > $O/gputests_sb1.log 2>&1; echo "sb1 gpu tests rc=$?"; tail -n 2 $O/gputests_sb1.log
for i in 1 2 3; do
  for v in base sh0 sh1 bg1 sb1; do
    export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
