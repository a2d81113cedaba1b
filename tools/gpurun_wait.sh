#!/bin/bash
# Development helper (build container only): submit ONE gpurun call, waiting while no GPU slot is free (gpurun exit 3 =
# nothing ran, nothing charged).  A call that ran -- whatever its result -- is never repeated.
# usage: tools/gpurun_wait.sh <timeout s> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
