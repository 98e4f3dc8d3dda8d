cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4t; mkdir -p $O
for v in 1 0; do
RZ_GLASS_CLAIMS=$v timeout -k 10 600 python3 profiles/scripts/pmc_collect.py $O/pmc_c2g_claims$v.json "rz_render_samples" -- python3 profiles/scripts/one_frame.py c2g > $O/pmc_c2g_$v.log 2>&1
done
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/r4t/pmc_c2g_claims1.json')); b=json.load(open('gpurun_out/r4t/pmc_c2g_claims0.json'))
for k in sorted(a):
    if k.startswith('_') or k not in b or not b[k]: continue
    r=a[k]/b[k]
    if abs(r-1)>0.04 or k in ('SQ_INSTS_VALU','SQ_WAVE_CYCLES','SQ_THREAD_CYCLES_VALU','SQ_ACTIVE_INST_VALU'): print(f"{k:34s} claims {a[k]:16.0f} group {b[k]:16.0f} ratio {r:.3f}")
print(a['_dispatch']); print(b['_dispatch'])
PY
