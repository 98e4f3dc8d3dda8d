#!/bin/bash
# Second call of round 4's collection: the bench line (with the CPU baseline over the whole frame) and the bench_configs table,
# both reading the PMC files that install_profiles4.py has put under profiles/r04_*.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r04; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 --cpu-full-frame > $O/bench_c2.json 2> $O/bench_c2.err
# RayZen's own scene at 1080p / 64 spp (added late in the round: its trace and counters are taken here, and copied to where
# bench_configs.py looks for them on this box; install_profiles4.py does the same at home)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ref64 -o run -- python3 profiles/scripts/one_frame.py ref64 > $O/kt_ref64.log 2>&1
python3 profiles/scripts/pmc_collect.py $O/pmc_ref64.json "rz_render_samples" -- python3 profiles/scripts/one_frame.py ref64 > $O/pmc_ref64.log 2>&1
mkdir -p profiles/r04_ref64 && cp $O/pmc_ref64.json profiles/r04_ref64/pmc_rz_render_samples.json
python3 bench_configs.py c1 ref ref16 ref64 c2close c2g glassbunny c3 c4 c4d c5 c5d > $O/bench_configs.log 2>&1
tail -3 $O/bench_configs.log
cat $O/bench_c2.json | head -c 1500
