# round 4: wait slots (per-group deferral, bounded scratch) -- parity, then A/B against the round's first commit ("base")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -15 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for v in base new; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so; fi
    timeout -k 10 400 python3 profiles/scripts/config_ms.py c2 c4 c3 c5full c2close >> $O/config_ms.log 2>&1 || exit 1
  done
done
unset RAYZEN_HIP_SO
cat $O/config_ms.log
