cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4g; mkdir -p $O; rm -f $O/prof.log
for v in ${VARIANTS:-baseprof prof}; do
  echo "===== $v" >> $O/prof.log
  RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so timeout -k 10 300 python3 profiles/scripts/prof_frame.py ${CFG:-c2} >> $O/prof.log 2>&1
done
grep -v "^\[rz_prof\] round" $O/prof.log | cut -c1-330
