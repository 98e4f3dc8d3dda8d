cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4f; mkdir -p $O
for rep in 1 2; do
  for v in base new; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so; fi
    timeout -k 10 400 python3 profiles/scripts/config_ms.py ${CFGS:-c2 c4 c3} >> $O/config_ms.log 2>&1 || exit 1
  done
done
unset RAYZEN_HIP_SO
cat $O/config_ms.log
