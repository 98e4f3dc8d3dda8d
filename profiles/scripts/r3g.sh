cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3v; mkdir -p $O
export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_prof.so
timeout -k 10 300 python3 profiles/scripts/prof_frame.py c2 > $O/prof_c2.log 2>&1
timeout -k 10 300 python3 profiles/scripts/prof_frame.py c4 > $O/prof_c4.log 2>&1
grep "round\|pool_trace\|compacting\|B phase\|refill\|blas loop\|leaf branch" $O/prof_c2.log $O/prof_c4.log | cut -c1-330
