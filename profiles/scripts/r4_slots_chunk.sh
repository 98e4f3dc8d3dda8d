# wait slots x pool chunk on the final build: PAIRS="32:256 16:256 ..." CFGS="c2 c4 ..."
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for p in ${PAIRS:-32:256 16:256 32:128 16:128}; do
  ns=${p%%:*}; ch=${p##*:}
  echo "== NS=$ns CHUNK=$ch"
  RZ_WAIT_SLOTS=$ns RZ_WPOOL_CHUNK=$ch timeout -k 10 400 python3 profiles/scripts/config_ms.py ${CFGS:-c2 c4 c5 c2g c3} || exit 1
done
done
