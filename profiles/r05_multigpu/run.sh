# round 5: the product with the IEEE slow paths out of line against the same + the generic-octant walk and Moeller-Trumbore's IEEE
# reciprocal behind calls (cold2); the whole GPU suite on the product; the N > 1 bench paths rehearsed on the one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
L=$PWD/rayzen_amd/lib
for i in 1 2 3; do
  for v in new cold2; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
unset RAYZEN_HIP_SO
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gputests.log
timeout -k 10 300 python bench.py --gpus 2 --loopback --steps 5 --warmup 2 > $O/bench_loop2.json 2> $O/bench_loop2.err; echo "loop2 rc=$?"
timeout -k 10 300 python bench.py --gpus 8 --loopback --steps 3 --warmup 1 > $O/bench_loop8.json 2> $O/bench_loop8.err; echo "loop8 rc=$?"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --reduce torch-gloo > $O/bench_ranks2_gloo.json 2> $O/bench_ranks2_gloo.err; echo "ranks2-gloo rc=$?"
