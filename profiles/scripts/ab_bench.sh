#!/bin/bash
# Same-box A/B: ab_bench.sh ROUNDS name1 name2 ... ("new" = the product library); prints kernel_ms per run.
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    if [ $v = new ]; then unset RAYZEN_HIP_SO; else export RAYZEN_HIP_SO=$PWD/rayzen_amd/lib/librayzen_hip_$v.so; fi
    python bench.py --no-cpu-baseline --steps 5 --warmup 2 ${AB_ARGS} | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['roofline']['kernel_ms'], d['ms_per_step'], flush=True)"
  done
done
