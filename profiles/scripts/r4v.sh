cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4v; mkdir -p $O
for cfg in c2 c2hidden; do
timeout -k 10 600 python3 profiles/scripts/pmc_collect.py $O/pmc_$cfg.json "rz_render_samples" -- python3 profiles/scripts/one_frame.py $cfg > $O/pmc_$cfg.log 2>&1
done
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/r4v/pmc_c2hidden.json')); b=json.load(open('gpurun_out/r4v/pmc_c2.json'))
for k in sorted(a):
    if k.startswith('_') or k not in b or not b[k]: continue
    r=a[k]/b[k]
    if abs(r-1)>0.03 or k in ('SQ_INSTS_VALU','SQ_WAVE_CYCLES','SQ_INSTS_SALU'): print(f"{k:34s} hidden {a[k]:16.0f} opaque {b[k]:16.0f} ratio {r:.3f}")
print(a['_dispatch']); print(b['_dispatch'])
PY
