cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3an; mkdir -p $O
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-full-frame > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench.json | cut -c1-300
