cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3ak; mkdir -p $O
RZ_FUZZ_CLAIM_SEEDS=400 timeout -k 10 1000 python3 -m pytest tests/test_fuzz_gpu.py -m gpu -x -q -k "claims_and_cross" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log | cut -c1-400; exit $rc
