cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3al; mkdir -p $O
timeout -k 10 300 python3 profiles/scripts/config_ms.py c2g glassbunny ref c2 > $O/ms.log 2>&1 || { cat $O/ms.log; exit 1; }
cat $O/ms.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "glass or transp or ior or random_scene or ref" > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log | cut -c1-300; exit $rc
