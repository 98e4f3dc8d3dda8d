cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for gc in 1 0; do
  echo "RZ_GLASS_CLAIMS=$gc"
  RZ_GLASS_CLAIMS=$gc timeout -k 10 300 python3 profiles/scripts/config_ms.py ref ref16 ref64 || exit 1
done
