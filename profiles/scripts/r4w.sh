cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4w; mkdir -p $O; rm -f $O/config_ms.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -4 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for v in 1 0; do
  echo "== RZ_LDS_UNIT_SUMS=$v" >> $O/config_ms.log
  RZ_LDS_UNIT_SUMS=$v timeout -k 10 300 python3 profiles/scripts/config_ms.py c2 c4 c2g c2close >> $O/config_ms.log 2>&1 || exit 1
done; done
cat $O/config_ms.log
