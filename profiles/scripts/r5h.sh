# round 5: the bench_configs table on the final build + where C4's and C2's wave cycles go (RZ_PROF build of the same sources)
bash profiles/scripts/collect_round5.sh 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r05
for c in c4 c2 ref64; do RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_prof.so timeout -k 10 300 python profiles/scripts/prof_frame.py $c > $O/prof_$c.log 2>&1; done
echo done
