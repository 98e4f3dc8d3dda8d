cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p; mkdir -p $O
for cfg in c2g glassbunny; do
RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_gstats.so timeout -k 10 300 python3 profiles/scripts/one_frame.py $cfg 2>&1 | tail -3
done
