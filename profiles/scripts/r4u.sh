cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for h in 1 0; do echo "== RZ_GLASS_BOX_HINT=$h"; RZ_GLASS_BOX_HINT=$h timeout -k 10 300 python3 profiles/scripts/glass_overhead.py; RZ_GLASS_BOX_HINT=$h timeout -k 10 300 python3 profiles/scripts/config_ms.py c2g glassbunny; done
