cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3aq; mkdir -p $O
for e in RZ_BLAS_STACK_WINDOW=3 RZ_COMPACT=0 RZ_CROSS_CLAIM_POOL=0 RZ_SMALL_SPP_CLAIMS=0 RZ_DEBUG_NO_POOL_MEMORY=1 RZ_WPOOL_CHUNK=40 RZ_HOST_RELAYOUT=1; do
  echo "== $e" >> $O/soak.log
  env $e timeout -k 10 500 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 >> $O/soak.log
done
cat $O/soak.log
