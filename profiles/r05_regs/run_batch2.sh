# round 5, batch 2: five single-switch variants against the product, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
L=$PWD/rayzen_amd/lib
for i in 1 2 3; do
  for v in new ktab hemi sl12 nt kptr; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c4 c5 c2g ref64 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
echo done
