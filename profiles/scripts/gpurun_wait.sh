#!/bin/bash
# gpurun_wait.sh LOG -- <command>: one gpurun call; when no GPU slot is free (exit 3: nothing ran, nothing charged) wait and ask again.
# A call that RAN is never repeated, whatever its exit code.
LOG=$1; shift; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout 1200 -- "$@" > "$LOG" 2>&1; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
