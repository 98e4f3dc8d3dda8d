# round 5, first GPU call: the whole GPU suite on the new build, the bench line, the N > 1 bench paths rehearsed on one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/gputests.log | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --gpus 2 --loopback --steps 5 --warmup 2 > $O/bench_loop2.json 2> $O/bench_loop2.err; echo "loop2 rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --gpus 8 --loopback --steps 3 --warmup 1 > $O/bench_loop8.json 2> $O/bench_loop8.err; echo "loop8 rc=$?" | tee -a $O/summary.txt
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --reduce torch-gloo > $O/bench_ranks2_gloo.json 2> $O/bench_ranks2_gloo.err; echo "ranks2-gloo rc=$?" | tee -a $O/summary.txt
for c in ref ref16 ref64 c4; do timeout -k 10 300 python bench_configs.py $c >> $O/configs.log 2>&1; done; tail -5 $O/configs.log | cut -c1-400
