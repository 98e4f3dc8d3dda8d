set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2 c2g c5full c4"
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_nospread.so
run A=1
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_base.so
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_nospread.so
run A=1
export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_prof.so
RZ_SMALL_SPP_CLAIMS=0 timeout -k 10 300 python3 profiles/scripts/prof_frame.py c4 > $O/prof_c4_noclaims.log 2>&1
RZ_SMALL_SPP_CLAIMS=0 RZ_SPREAD_MIN_INSTANCES=0 timeout -k 10 300 python3 profiles/scripts/prof_frame.py c4 > $O/prof_c4_noclaims_nospread.log 2>&1
timeout -k 10 300 python3 profiles/scripts/prof_frame.py c4 > $O/prof_c4_claims.log 2>&1
timeout -k 10 300 python3 profiles/scripts/prof_frame.py ref > $O/prof_ref.log 2>&1
cat $O/ab.log
