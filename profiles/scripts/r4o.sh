# GLASS claims: parity (the whole GPU suite), then timings of the transparent workloads with and without them
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4o; mkdir -p $O; rm -f $O/config_ms.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -15 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for v in 1 0; do
  echo "== RZ_GLASS_CLAIMS=$v" >> $O/config_ms.log
  RZ_GLASS_CLAIMS=$v timeout -k 10 300 python3 profiles/scripts/config_ms.py c2g glassbunny c2 >> $O/config_ms.log 2>&1 || exit 1
done; done
cat $O/config_ms.log
RZ_GLASS_CLAIMS=1 timeout -k 10 300 python3 profiles/scripts/glass_overhead.py
