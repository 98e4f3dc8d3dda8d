cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m; mkdir -p $O; rm -f $O/config_ms.log
for ch in 256 384 512; do for ns in 32 48; do
  echo "== CHUNK=$ch NS=$ns" >> $O/config_ms.log
  RZ_WPOOL_CHUNK=$ch RZ_WAIT_SLOTS=$ns timeout -k 10 300 python3 profiles/scripts/config_ms.py c2 c4 >> $O/config_ms.log 2>&1
done; done
cat $O/config_ms.log
