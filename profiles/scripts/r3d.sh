set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
true
true
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2 c4 c2g ref c5 c5full"
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run A=1
run RZ_SPREAD_MIN_INSTANCES=0
CFG="c2 c5 c5full"
run RZ_SPREAD_MIN_INSTANCES=2
run RZ_CLAIM_RUN=1
run RZ_CLAIM_RUN=2
CFG="c4"
run RZ_SMALL_SPP_CLAIMS=0
for pc in 2 4 8 16; do for rn in 1 2; do run RZ_GROUPS_PER_CLAIM=$pc RZ_CLAIM_RUN=$rn; done; done
run RZ_GROUPS_PER_CLAIM=8 RZ_CLAIM_RUN=1 RZ_SPREAD_MIN_INSTANCES=0
run RZ_GROUPS_PER_CLAIM=8 RZ_CLAIM_RUN=8
CFG="c2 c4 c2g ref c5 c5full"
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run A=1
cat $O/ab.log
