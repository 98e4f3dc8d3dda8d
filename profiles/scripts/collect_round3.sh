#!/bin/bash
# Collects round 3's profile evidence on the GPU box into gpurun_out/prof_r03/ (install_profiles.py copies what is to be judged
# into profiles/r03_*): the bench line, kernel trace + stats and PMC passes (one counter group per pass, never combined with a
# trace domain) for C2 (the bench workload), C4, C5 at its stated 128 spp, the glass + mirror frame and RayZen's own workload.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 --cpu-full-frame > $O/bench_c2.json 2> $O/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -o run -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/kt_c2.log 2>&1
python3 profiles/scripts/pmc_collect.py $O/pmc_c2.json "rz_render_samples" --workload 1920,1080,64,4,76 -- python3 profiles/scripts/one_frame.py c2 > $O/pmc_c2.log 2>&1
for c in c4 c5full c2g ref; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o run -- python3 profiles/scripts/one_frame.py $c > $O/kt_$c.log 2>&1
  python3 profiles/scripts/pmc_collect.py $O/pmc_$c.json "rz_render_samples" -- python3 profiles/scripts/one_frame.py $c > $O/pmc_$c.log 2>&1
done
python3 bench_configs.py c1 ref c4 c4d c5 c5d > $O/bench_configs.log 2>&1
python3 profiles/scripts/rank_share.py 1 2 4 8 > $O/rank_share.log 2>&1
ls $O
