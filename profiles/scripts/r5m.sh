# round 5: where the wave cycles of the NON-counting transparent (and opaque) claims kernels go: RZ_GSTATS build with phase timers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p $O
for c in c2g glassbunny c2hidden ref64 c2 c4; do RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_gst.so timeout -k 10 300 python profiles/scripts/one_frame.py $c > $O/gst_$c.log 2>&1; grep rz_gstats $O/gst_$c.log | tail -2 | cut -c1-400; done
