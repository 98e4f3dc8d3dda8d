# round 5, second session: with the shared sin / cos core behind a call, does the hemisphere draw want a call of its own after all?
# hg1: in the transparent kernels; ho1: in the opaque ones; sh2 = the tree's default (inline in both).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5u; mkdir -p $O
L=$PWD/rayzen_amd/lib
for i in 1 2 3; do
  for v in sh2 hg1 ho1; do
    export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref16 ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
