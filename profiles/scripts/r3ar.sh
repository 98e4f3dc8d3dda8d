cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ar; mkdir -p $O
for e in RZ_BLAS_STACK_WINDOW=3 RZ_COMPACT=0 RZ_SMALL_SPP_CLAIMS=0 RZ_DEBUG_NO_POOL_MEMORY=1; do
  echo "== $e" >> $O/soak.log
  env $e timeout -k 10 500 python3 -m pytest tests/test_gpu_configs_full.py tests/test_fuzz_gpu.py -m gpu -q --tb=line -k "not random_scene" 2>&1 | grep -E "^/|Error|passed|failed" | cut -c1-300 | sort | uniq -c | sort -rn | head -12 >> $O/soak.log
done
cat $O/soak.log
