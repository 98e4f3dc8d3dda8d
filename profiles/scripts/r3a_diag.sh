set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p $O
python3 profiles/scripts/config_ms.py c2 c4 c2g glassbunny ref > $O/config_ms.log 2>&1 &&
for w in c4 ref c2g c2; do
  RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_prof.so timeout -k 10 300 python3 profiles/scripts/prof_frame.py $w > $O/prof_$w.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2g -o run -- python3 profiles/scripts/one_frame.py c2g > $O/kt_c2g.log 2>&1 &&
python3 profiles/scripts/pmc_collect.py $O/pmc_c2g.json rz_render_samples -- python3 profiles/scripts/one_frame.py c2g > $O/pmc_c2g.log 2>&1 &&
python3 profiles/scripts/pmc_collect.py $O/pmc_ref.json rz_render_samples -- python3 profiles/scripts/one_frame.py ref > $O/pmc_ref.log 2>&1
cat $O/config_ms.log
timeout -k 10 400 python3 -m pytest tests/test_gpu_configs_full.py -q -m gpu -k c5 > $O/c5_tests.log 2>&1; tail -3 $O/c5_tests.log
