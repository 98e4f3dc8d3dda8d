cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ah; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_lanes profiles/scripts/ta_lanes.hip > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
timeout -k 10 120 /tmp/ta_lanes > $O/ta_lanes.txt 2>&1 || { cat $O/ta_lanes.txt; exit 1; }
cat $O/ta_lanes.txt
