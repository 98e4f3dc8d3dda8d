cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3af; mkdir -p $O
for v in "" _cs64 _cs256; do
  echo "== lib$v" >> $O/modes.log
  RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip$v.so RZ_GLASS_DEBUG=1 timeout -k 10 400 python3 profiles/scripts/glass_modes.py c2g glassbunny c2g16 >> $O/modes.log 2>&1 || { cat $O/modes.log; exit 1; }
done
grep -v "^\[rz_glass\]" $O/modes.log
grep "^\[rz_glass\]\|==" $O/modes.log | awk 'NR%1==0' | cut -c1-120 > $O/passes.log
