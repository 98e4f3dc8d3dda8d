# round 5, second session: the final tree -- whole GPU suite on the product (math flavour 1), the same suite on the flavour-0 build of
# the same sources (goldens excluded: they hold flavour-1 frames), smoke, a 3 000-scene soak of the three fuzz families on the product
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5v; mkdir -p $O
L=$PWD/rayzen_amd/lib
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 2 $O/gputests.log
RAYZEN_HIP_SO=$L/librayzen_hip_f0.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_f0.log 2>&1; echo "f0 gpu tests rc=$?"; tail -n 2 $O/gputests_f0.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
RZ_FUZZ_GLASS_CLAIM_SEEDS=1500 RZ_FUZZ_CLAIM_SEEDS=800 RZ_FUZZ_SEEDS=700 timeout -k 10 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > $O/soak.log 2>&1; echo "soak rc=$?"; tail -n 2 $O/soak.log
