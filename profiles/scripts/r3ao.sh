cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ao; mkdir -p $O
RZ_FUZZ_SEEDS=1200 RZ_FUZZ_CLAIM_SEEDS=2000 timeout -k 10 1100 python3 -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-400; exit $rc
