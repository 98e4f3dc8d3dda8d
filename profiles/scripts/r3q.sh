cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rayzen_amd/csrc/hip -I include -o /tmp/div_sqrt_proof profiles/scripts/div_sqrt_proof.hip 2> $O/build.err
timeout -k 10 500 /tmp/div_sqrt_proof > $O/proof.txt 2>&1; echo "rc=$?"
cat $O/proof.txt
