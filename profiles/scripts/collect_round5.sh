#!/bin/bash
# Collects round 5's profile evidence on the GPU box into gpurun_out/prof_r05/ (install_profiles5.py copies what is to be judged
# into profiles/r05_*).  Usage: collect_round5.sh PART   (1: C2 + the bench line + rank shares; 2: c2close c3 c4 ref ref16; 3: c5full c2g glassbunny ref64;
# 4: the bench_configs table, reading the PMC files the earlier parts have left under profiles/r05_*)
# Per configuration: kernel trace + stats, PMC passes (one counter group per pass, never combined with a trace domain) and the
# launch's algorithmic tallies (rz_render_counted) for tests/test_workmodel.py.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r05; mkdir -p $O
one() {
  c=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o run -- python3 profiles/scripts/one_frame.py $c > $O/kt_$c.log 2>&1
  python3 profiles/scripts/pmc_collect.py $O/pmc_$c.json "rz_render_samples" "$@" -- python3 profiles/scripts/one_frame.py $c > $O/pmc_$c.log 2>&1
  python3 profiles/scripts/one_frame.py $c --counters $O/counters_$c.json > $O/counters_$c.log 2>&1
}
case "$1" in
1)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -o run -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/kt_bench.log 2>&1
  one c2 --workload 1920,1080,64,4,76
  mkdir -p profiles/r05_c2_kernel && cp $O/pmc_c2.json profiles/r05_c2_kernel/pmc_rz_render_samples.json
  python3 bench.py --steps 20 --warmup 5 --cpu-full-frame > $O/bench_c2.json 2> $O/bench_c2.err
  python3 profiles/scripts/rank_share.py 1 2 4 8 > $O/rank_share.log 2>&1
  ;;
2) for c in c2close c3 c4 ref ref16; do one $c; done ;;
3) for c in c5full c2g glassbunny ref64; do one $c; done ;;
4)
  for c in c2close c3 c4 ref ref16 c2g glassbunny ref64; do [ -f $O/pmc_$c.json ] && mkdir -p profiles/r05_$c && cp $O/pmc_$c.json profiles/r05_$c/pmc_rz_render_samples.json; done
  [ -f $O/pmc_c5full.json ] && mkdir -p profiles/r05_c5 && cp $O/pmc_c5full.json profiles/r05_c5/pmc_rz_render_samples.json
  python3 bench_configs.py c1 ref ref16 ref64 c2close c2g glassbunny c3 c4 c4d c5 c5d > $O/bench_configs.log 2>&1
  tail -3 $O/bench_configs.log | cut -c1-300
  ;;
esac
ls $O | head -50
