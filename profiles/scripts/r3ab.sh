cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ab; mkdir -p $O
timeout -k 10 600 python3 profiles/scripts/glass_modes.py c2g glassbunny c2g16 c2g256 > $O/modes.log 2>&1 || { cat $O/modes.log; exit 1; }
cat $O/modes.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "glass or transp or ior or fuzz" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
