# usage: r3_ab.sh OUTDIR "configs" variant...   (variant "new" = the product library): config_ms.py per variant, twice
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O; CFG=$2; shift; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so; fi
    timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/config_ms.log 2>&1 || exit 1
  done
done
unset RAYZEN_HIP_SO
cat $O/config_ms.log
