cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4r; mkdir -p $O; rm -f $O/config_ms.log
echo "== default" >> $O/config_ms.log; timeout -k 10 200 python3 profiles/scripts/config_ms.py ref >> $O/config_ms.log 2>&1
for pc in 1 2 4 8; do for run in 1 $pc; do
  echo "== RZ_GROUPS_PER_CLAIM=$pc RZ_CLAIM_RUN=$run" >> $O/config_ms.log
  RZ_GROUPS_PER_CLAIM=$pc RZ_CLAIM_RUN=$run timeout -k 10 200 python3 profiles/scripts/config_ms.py ref >> $O/config_ms.log 2>&1
done; done
echo "== RZ_GROUPS_PER_CLAIM=4 RZ_GLASS_CLAIMS=0" >> $O/config_ms.log; RZ_GROUPS_PER_CLAIM=4 RZ_GLASS_CLAIMS=0 timeout -k 10 200 python3 profiles/scripts/config_ms.py ref >> $O/config_ms.log 2>&1
cat $O/config_ms.log
