cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ag; mkdir -p $O
RZ_GLASS_DEBUG=1 timeout -k 10 400 python3 profiles/scripts/glass_modes.py c2g glassbunny c2g16 >> $O/modes.log 2>&1 || { cat $O/modes.log; exit 1; }
grep -v "^\[rz_glass\]" $O/modes.log
