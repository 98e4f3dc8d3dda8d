#!/bin/bash
# Collects the round's profile evidence on the GPU box into gpurun_out/prof_r02/ (copy what is to be judged into profiles/):
#   kernel trace + stats of bench.py, PMC passes for C2 / C4 / C5 frames, the issue microbenchmark and its own counters.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02b; mkdir -p $O
python3 bench.py --steps 10 --warmup 2 --cpu-full-frame > $O/bench_c2.json 2> $O/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -o run -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/kt_c2.log 2>&1
python3 profiles/scripts/pmc_collect.py $O/pmc_c2.json rz_render_samples --workload 1920,1080,64,4,76 -- python3 profiles/scripts/one_frame.py c2 > $O/pmc_c2.log 2>&1
for c in c4 c5full; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -o run -- python3 profiles/scripts/one_frame.py $c > $O/kt_$c.log 2>&1
  python3 profiles/scripts/pmc_collect.py $O/pmc_$c.json rz_render_samples -- python3 profiles/scripts/one_frame.py $c > $O/pmc_$c.log 2>&1
done
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/valu_issue profiles/scripts/valu_issue.hip 2>/dev/null
timeout -k 10 600 /tmp/valu_issue > $O/valu_issue.txt 2>&1
# what the SQ counters read while one instruction class saturates the SIMDs (4 waves per SIMD): v_fma_f32 (op 0), v_fma_f64 (op 32)
for op in 0 32; do
  python3 profiles/scripts/pmc_collect.py $O/pmc_valu_issue_op$op.json "k<$op>" --groups "SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_ANY,SQ_INSTS_SALU,GRBM_GUI_ACTIVE" -- /tmp/valu_issue $op 4 > $O/pmc_valu_issue_op$op.log 2>&1
done
ls $O
