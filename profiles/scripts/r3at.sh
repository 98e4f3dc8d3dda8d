cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3at; mkdir -p $O
timeout -k 10 300 python3 profiles/scripts/config_ms.py c2 c4 c5 c5full c3 ref c2g > $O/ms.log 2>&1 || { cat $O/ms.log; exit 1; }
timeout -k 10 300 python3 profiles/scripts/config_ms.py c2 c4 c5 c5full c3 ref c2g >> $O/ms.log 2>&1 || { cat $O/ms.log; exit 1; }
cat $O/ms.log
