cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 8 4; do
echo "== default"; python3 profiles/scripts/share_sweep.py $n
for pc in 1 2 8; do echo "== RZ_GROUPS_PER_CLAIM=$pc"; RZ_GROUPS_PER_CLAIM=$pc python3 profiles/scripts/share_sweep.py $n; done
for ch in 64 128; do echo "== RZ_WPOOL_CHUNK=$ch"; RZ_WPOOL_CHUNK=$ch python3 profiles/scripts/share_sweep.py $n; done
echo "== RZ_CROSS_CLAIM_POOL=0"; RZ_CROSS_CLAIM_POOL=0 python3 profiles/scripts/share_sweep.py $n
echo "== RZ_COMPACT=0"; RZ_COMPACT=0 python3 profiles/scripts/share_sweep.py $n
echo "== RZ_GROUPS_PER_CLAIM=2 RZ_WPOOL_CHUNK=64"; RZ_GROUPS_PER_CLAIM=2 RZ_WPOOL_CHUNK=64 python3 profiles/scripts/share_sweep.py $n
done
