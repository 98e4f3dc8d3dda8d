#!/bin/bash
# Build librayzen_hip at git revision $1 (or WORK for the working tree) into rayzen_amd/lib/librayzen_hip_$2.so with
# extra hipcc flags $3... (for same-box A/B runs: RAYZEN_HIP_SO=rayzen_amd/lib/librayzen_hip_$2.so python bench.py ...).
set -e
REV=${1:-HEAD}; NAME=${2:-base}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$REV" = WORK ]; then
  H=$ROOT/rayzen_amd/csrc/hip; INC=$ROOT/include
else
  T=$ROOT/gpurun_out/src_$NAME
  rm -rf "$T"; mkdir -p "$T/hip" "$T/include"
  for f in $(git -C "$ROOT" ls-tree --name-only "$REV" rayzen_amd/csrc/hip/); do git -C "$ROOT" show "$REV:$f" > "$T/hip/$(basename $f)"; done
  for f in $(git -C "$ROOT" ls-tree --name-only "$REV" include/); do git -C "$ROOT" show "$REV:$f" > "$T/include/$(basename $f)"; done
  H=$T/hip; INC=$T/include
fi
# the stamp rz_source_hash() returns: the tree's hash over sources + the product's flags + these extra flags for WORK builds
STAMP=$(cd "$ROOT" && python3 -c "import sys; from rayzen_amd import build; print(build.source_hash(tuple(sys.argv[1:])))" "$@")
[ "$REV" = WORK ] || STAMP=$(printf '%064d' 0)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -greedy-regclass-priority-trumps-globalness "$@" "-DRZ_SOURCE_HASH=\"$STAMP\"" -I "$INC" -I "$H" \
    -shared -o "$ROOT/rayzen_amd/lib/librayzen_hip_$NAME.so" "$H"/*.hip -ldl 2> "$ROOT/gpurun_out/ab_build_$NAME.err" || { grep -m5 "error" "$ROOT/gpurun_out/ab_build_$NAME.err"; echo "BUILD FAILED: librayzen_hip_$NAME.so"; exit 1; }
echo "built librayzen_hip_$NAME.so from $REV $*"
