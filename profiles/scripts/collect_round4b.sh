#!/bin/bash
# Second call of round 4's collection: the bench line (with the CPU baseline over the whole frame) and the bench_configs table,
# both reading the PMC files that install_profiles4.py has put under profiles/r04_*.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r04; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 --cpu-full-frame > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench_configs.py c1 ref c2close c2g glassbunny c3 c4 c4d c5 c5d > $O/bench_configs.log 2>&1
tail -3 $O/bench_configs.log
cat $O/bench_c2.json | head -c 1500
