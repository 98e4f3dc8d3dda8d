# RZ_PARK_KEEP sweep (a unit parks its late paths only once fewer than this many are left): LIB=<variant> KEEPS="65 48 32" CFGS="..."
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$LIB.so
for k in ${KEEPS:-65 48 32 16}; do
  echo "RZ_PARK_KEEP=$k"
  RZ_PARK_KEEP=$k timeout -k 10 400 python3 profiles/scripts/config_ms.py ${CFGS:-c2 c2close c4 c2g ref16 ref64 c5} || exit 1
done
