cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3aw; mkdir -p $O
for i in 1 2 3; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['work_model']['frac'])" >> $O/bench3.log || exit 1; done
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
cat $O/bench3.log; cut -c1-200 $O/bench_default.json
