# round 5, second session: flavour-1 variants on the same box: the hemisphere draw inline (f1i), + rand behind a call (f1r), + the shared
# sin / cos core behind a call (f1s), against the product (binary64 built-ins)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5q; mkdir -p $O
L=$PWD/rayzen_amd/lib
for i in 1 2 3; do
  for v in new f1i f1r f1s; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
