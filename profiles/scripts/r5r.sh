# round 5, second session: the new default build (math flavour 1) -- the whole GPU suite, the flavour-0 build through the same suite
# (goldens excluded: they hold flavour-1 frames now), then the bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5r; mkdir -p $O
L=$PWD/rayzen_amd/lib
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/gputests.log
RAYZEN_HIP_SO=$L/librayzen_hip_f0.so timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not golden" > $O/gputests_f0.log 2>&1; echo "f0 gpu tests rc=$?"; tail -n 3 $O/gputests_f0.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
