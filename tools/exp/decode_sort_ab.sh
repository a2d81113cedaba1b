#!/bin/bash
# k_decode_sort at configs[3] (10 000 sketch files): counting inside buckets vs the bitonic network for every sketch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in default network; do
  rm -rf /tmp/kts
  if [ $v = default ]; then timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -o r -- python3 $R/tools/exp/c4_files.py 1 > /dev/null 2>&1
  else SPSP_DEBUG_DECODE_SORT=network timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -o r -- python3 $R/tools/exp/c4_files.py 1 > /dev/null 2>&1; fi
  python3 $R/tools/prof_summary.py $(find /tmp/kts -name "*kernel_trace.csv") /tmp/kts/s.md > /dev/null
  echo "== $v"; grep "k_decode" /tmp/kts/s.md | cut -c1-150
done
