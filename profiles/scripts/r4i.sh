cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4i; mkdir -p $O
for ns in 16 20 24 28 32; do echo "== NS=$ns" >> $O/config_ms.log; RZ_WAIT_SLOTS=$ns timeout -k 10 200 python3 profiles/scripts/config_ms.py c2 c4 >> $O/config_ms.log 2>&1; done
cat $O/config_ms.log
