cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3aj; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench.json
