cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 profiles/scripts/pmc_collect.py $O/pmc_c2.json "rz_render_samples" --groups FETCH_SIZE,TCC_HIT_sum WRITE_SIZE,TCC_MISS_sum -- python3 profiles/scripts/one_frame.py c2 > $O/pmc_c2.log 2>&1
python3 -c "
import json; p=json.load(open('$O/pmc_c2.json')); print({k:p[k] for k in p if not k.startswith('_')}); print('traffic GB', (2*p['FETCH_SIZE']+p['WRITE_SIZE'])*1024/1e9)"
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from rayzen_amd import scene as S
from rayzen_amd.renderer import Renderer, frame_params
for name in ('c2','c4','c3','c5full'):
    sc,W,H,spp,b=S.named_config(name); r=Renderer(0); r.upload_scene(sc); r.set_frame(frame_params(sc.camera,W,H,len(sc.lights),b,spp)); r.render(); r.sync(); print(name, r.debug_last_plan(), flush=True); r.close()
PY
