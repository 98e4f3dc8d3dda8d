# round 5: the ordered sums at a claim's end (14 % of a C2 wave's lifetime by the RZ_GSTATS timers): loads in flight per chunk of the chain
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; mkdir -p $O
L=$PWD/rayzen_amd/lib
for i in 1 2 3; do
  for v in new sum4 sum16 sum32; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
