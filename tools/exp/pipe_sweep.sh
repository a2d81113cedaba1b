#!/bin/bash
# the 100 x 5 Mbp file pipeline at 16 threads for a few (batches in flight, batch budget) pairs: best and median wall of 7 runs each
R=${GRAFT_REPO_ROOT:-$(pwd)}
for slots in 4 6 8; do for mb in 8 16 32; do
  echo -n "slots $slots budget ${mb}MB: "
  SPSP_DEBUG_PIPE_SLOTS=$slots SPSP_DEBUG_PIPE_BUDGET_MB=$mb SPSP_DEBUG_PIPE_TRACE=1 timeout -k 10 300 python3 $R/tools/e2e_files.py 100 5000000 16 7 2>&1 >/dev/null | grep "pipe\] call" | tail -7 | awk '{print $3}' | sort -n | tr '\n' ' '
  echo
done; done
