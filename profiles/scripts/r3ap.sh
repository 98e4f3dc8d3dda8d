cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ap; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
bash profiles/scripts/collect_round3.sh > $O/collect.log 2>&1; tail -2 $O/collect.log
