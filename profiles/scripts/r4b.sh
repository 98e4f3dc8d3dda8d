# round 4: prefix snapshot of the speculating kernel -- parity tests, then A/B by RZ_GLASS_SNAPSHOT, then the variant's pure overhead
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -5 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for v in 1 0; do
    echo "== RZ_GLASS_SNAPSHOT=$v" >> $O/config_ms.log
    RZ_GLASS_SNAPSHOT=$v timeout -k 10 300 python3 profiles/scripts/config_ms.py c2g glassbunny >> $O/config_ms.log 2>&1 || exit 1
  done
done
cat $O/config_ms.log
timeout -k 10 300 python3 profiles/scripts/glass_overhead.py > $O/glass_overhead.log 2>&1
cat $O/glass_overhead.log
