# round 4: wait slots -- sensitivity to the number of slots and the pool chunk
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; mkdir -p $O
for ns in 32 48 64; do for ch in 128 256; do
  echo "== RZ_WAIT_SLOTS=$ns RZ_WPOOL_CHUNK=$ch" >> $O/config_ms.log
  RZ_WAIT_SLOTS=$ns RZ_WPOOL_CHUNK=$ch timeout -k 10 300 python3 profiles/scripts/config_ms.py c2 c4 >> $O/config_ms.log 2>&1 || exit 1
done; done
echo "== RZ_CROSS_CLAIM_POOL=0" >> $O/config_ms.log
RZ_CROSS_CLAIM_POOL=0 timeout -k 10 300 python3 profiles/scripts/config_ms.py c2 c4 >> $O/config_ms.log 2>&1
cat $O/config_ms.log
