# round 4: tests on the final flags, then PROF frames (pair statistics for the quantised pre-test question) of C2 and C4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4s; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for cfg in c2 c4; do
  RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_prof.so timeout -k 10 300 python3 profiles/scripts/prof_frame.py $cfg > $O/prof_$cfg.log 2>&1
  grep "descend steps of the general" $O/prof_$cfg.log
done
timeout -k 10 400 python3 profiles/scripts/config_ms.py c2 c2close c2g glassbunny c3 c4 c5full ref
