# round 5, batch 3: the adopted pair (one-wait 48-B scalar fetches + the hemisphere draw out of line in the opaque kernels) = the product,
# against the same without the out-of-line draw, + two further single-switch variants; then the whole GPU suite on the product
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
L=$PWD/rayzen_amd/lib
for i in 1 2 3; do
  for v in new hemiall randni tripf; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so; fi
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g ref64 2>&1 | tail -1 | tee -a $O/ab.log || exit 1
  done
done
unset RAYZEN_HIP_SO
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gputests.log
