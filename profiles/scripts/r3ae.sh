cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ae; mkdir -p $O
RZ_GLASS_DEBUG=1 timeout -k 10 300 python3 profiles/scripts/glass_modes.py c2g16 > $O/modes16.log 2>&1 || { cat $O/modes16.log; exit 1; }
cat $O/modes16.log
RZ_GLASS_DEBUG=1 timeout -k 10 600 python3 profiles/scripts/glass_modes.py c2g glassbunny c2g256 > $O/modes.log 2>&1 || { cat $O/modes.log; exit 1; }
cat $O/modes.log
