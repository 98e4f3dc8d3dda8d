#!/bin/bash
# min kernel ms over several contexts per library (placement makes single runs bimodal): mode_min.sh name...
for v in "$@"; do
  if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so; fi
  python profiles/scripts/mode_probe.py 2>/dev/null | python -c "
import sys,ast
v=[]
for l in sys.stdin:
    v+=ast.literal_eval(l.split(' ',1)[1])
v.sort(); print('$v', 'min', v[0], 'max', v[-1], flush=True)"
done
