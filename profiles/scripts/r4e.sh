cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 600 python3 profiles/scripts/pmc_collect.py $O/pmc_c2.json "rz_render_samples" -- python3 profiles/scripts/one_frame.py c2 > $O/pmc_c2.log 2>&1
python3 - <<'PY'
import json
n=json.load(open('gpurun_out/r4e/pmc_c2.json')); b=json.load(open('profiles/r03_c2_kernel/pmc_rz_render_samples.json'))
for k in sorted(n):
    if k.startswith('_'): continue
    print(f"{k:36s} new {n[k]:16.0f} base {b.get(k,float('nan')):16.0f} ratio {n[k]/b[k] if b.get(k) else float('nan'):.3f}")
print(n['_dispatch'])
PY
