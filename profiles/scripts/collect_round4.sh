#!/bin/bash
# Collects round 4's profile evidence on the GPU box into gpurun_out/prof_r04/ (install_profiles4.py copies what is to be judged
# into profiles/r04_*): kernel trace + stats and PMC passes (one counter group per pass, never combined with a trace domain; the
# instruction-cache counters among them) for C2 (the bench workload), C4, C5 at its stated 128 spp, the glass + mirror frame, the
# mesh of glass, the geometry-dominated companion of C2 and RayZen's own workload; the rank shares.  The bench line and the
# bench_configs table are taken in a SECOND call (collect_round4b.sh), when the PMC files are in profiles/ for them to read.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r04; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -o run -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/kt_c2.log 2>&1
python3 profiles/scripts/pmc_collect.py $O/pmc_c2.json "rz_render_samples" --workload 1920,1080,64,4,76 -- python3 profiles/scripts/one_frame.py c2 > $O/pmc_c2.log 2>&1
for c in c4 c5full c2g glassbunny c2close ref c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o run -- python3 profiles/scripts/one_frame.py $c > $O/kt_$c.log 2>&1
  python3 profiles/scripts/pmc_collect.py $O/pmc_$c.json "rz_render_samples" -- python3 profiles/scripts/one_frame.py $c > $O/pmc_$c.log 2>&1
done
python3 profiles/scripts/rank_share.py 1 2 4 8 > $O/rank_share.log 2>&1
ls $O
