# round 5, second session: sky units (dark, nothing parked, 64 spp, opaque) sum and store their pixel on the spot through LDS (sk1 = the tree default)
# against the same sources without (sk0): parity on the whole GPU suite, same-box timings, HBM counters of C2 and C5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5ah; mkdir -p $O
L=$PWD/rayzen_amd/lib
RAYZEN_HIP_SO=$L/librayzen_hip_sk1.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_sk1.log 2>&1; echo "sk1 gpu tests rc=$?"; tail -n 2 $O/gputests_sk1.log
for i in 1 2 3; do
  for v in sk0 sk1; do
    export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
    timeout -k 10 300 python profiles/scripts/config_ms.py c2 c2close c3 c4 c5 c2g glassbunny ref ref16 ref64 2>&1 | tail -n 1 | tee -a $O/ab.log || exit 1
  done
done
for v in sk0 sk1; do
  export RAYZEN_HIP_SO=$L/librayzen_hip_$v.so
  for c in c2 c5full; do
    python3 profiles/scripts/pmc_collect.py $O/pmc_${c}_$v.json "rz_render_samples" --groups FETCH_SIZE,TCC_HIT_sum WRITE_SIZE,TCC_MISS_sum -- python3 profiles/scripts/one_frame.py $c > $O/pmc_${c}_$v.log 2>&1
  done
done
python3 - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5ah/pmc_*.json')):
    p = json.load(open(f)); print(f, 'HBM GB %.2f' % ((2 * p['FETCH_SIZE'] + p['WRITE_SIZE']) * 1024 / 1e9), 'fetch', p['FETCH_SIZE'], 'write', p['WRITE_SIZE'], 'ms', p['_dispatch']['duration_ns_under_profiler'] / 1e6)
P
