cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4j; mkdir -p $O
for pad in 0 96 160 352 1056; do echo "== PAD=$pad" >> $O/config_ms.log; RZ_SLOT_STRIDE_PAD=$pad timeout -k 10 200 python3 profiles/scripts/config_ms.py c2 >> $O/config_ms.log 2>&1; done
cat $O/config_ms.log
