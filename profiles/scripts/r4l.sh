# A/B of library variants: VARS="base u16 u32" CFGS="c2 c4"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l; mkdir -p $O; rm -f $O/config_ms.log
for rep in 1 2; do
  for v in $VARS; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so; fi
    timeout -k 10 400 python3 profiles/scripts/config_ms.py ${CFGS:-c2 c4} >> $O/config_ms.log 2>&1 || exit 1
  done
done
cat $O/config_ms.log
