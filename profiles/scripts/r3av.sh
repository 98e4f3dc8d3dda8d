cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3av; mkdir -p $O
run() { echo "== $*" >> $O/ab.log; env "$@" timeout -k 10 300 python3 profiles/scripts/config_ms.py $CFG >> $O/ab.log 2>&1 || exit 1; }
CFG="c2 c4 c5 c3 c2g"
run A=1
run RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_oct0.so
run A=1
cat $O/ab.log
