# round 5, second session: RZ_MATH_FLAVOUR=1 (llvmpipe's sin / cos / acos in binary32) against the product's binary64 built-ins:
# parity of the variant with the oracle in the same flavour (the suite's session fixture asks the loaded library: rz_math_flavour();
# whole GPU suite minus the oracle-made goldens, which hold flavour-0 frames), then same-box timings.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_f1.so timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not golden" > $O/gputests_f1.log 2>&1; echo "f1 gpu tests rc=$?"; tail -n 3 $O/gputests_f1.log
for i in 1 2 3; do
  for v in new f1 f1i; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
