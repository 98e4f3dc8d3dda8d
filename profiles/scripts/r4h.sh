cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p $O
for v in base nosums noatom; do
  RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so timeout -k 10 200 python3 profiles/scripts/config_ms.py c2 >> $O/config_ms.log 2>&1
done
for ns in 16 24; do echo "== NS=$ns" >> $O/config_ms.log; RZ_WAIT_SLOTS=$ns RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_noatom.so timeout -k 10 200 python3 profiles/scripts/config_ms.py c2 >> $O/config_ms.log 2>&1; done
cat $O/config_ms.log
