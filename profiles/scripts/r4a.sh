# round 4, first GPU call: which instruction-cache counters this rocprofv3 knows; the GPU suite on the round's first build (new
# tallies, the fixed three-numerator admission test); the round's starting table of kernel times; the I-cache pass on C2.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -i -E "icache|ifetch|SQC_" $O/avail.txt | head -60 > $O/avail_icache.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 profiles/scripts/config_ms.py c2 c2close c2g glassbunny c3 c4 c5full ref > $O/config_ms.log 2>&1 || exit 1
cat $O/config_ms.log
timeout -k 10 300 python3 profiles/scripts/pmc_collect.py $O/pmc_icache_c2.json "rz_render_samples" --groups SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES,SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH,SQ_IFETCH_LEVEL,SQ_WAIT_IFETCH SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_INSTS_VALU -- python3 profiles/scripts/one_frame.py c2 > $O/pmc_icache_c2.log 2>&1
cat $O/pmc_icache_c2.json | head -40
