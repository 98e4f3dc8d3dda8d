# round 5: where the transparent frames' wave cycles go (RZ_PROF + RZ_GSTATS build of the final sources), and a soak of the final build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
for c in c2g glassbunny c2hidden; do RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_profg.so timeout -k 10 300 python profiles/scripts/prof_frame.py $c > $O/prof_$c.log 2>&1; done
RZ_FUZZ_GLASS_CLAIM_SEEDS=6000 RZ_FUZZ_CLAIM_SEEDS=2500 RZ_FUZZ_SEEDS=1500 timeout -k 10 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > $O/soak.log 2>&1; echo "soak rc=$?"; tail -2 $O/soak.log
