#!/bin/bash
# GPU box: kernel-trace timelines of bench.py's step under the schedules named on the command line
# usage: tools/trace_schedules.sh single tail ...   -> gpurun_out/timeline_<schedule>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for s in "$@"; do
  rm -rf /tmp/kt_$s
  BENCH_SCHEDULE=$s timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$s -o r -- python3 $R/bench.py --experiment --no-cpu-baseline --no-extras > $R/gpurun_out/line_$s.json 2>/dev/null || exit 1
  python3 $R/tools/timeline_mid.py $(find /tmp/kt_$s -name "*kernel_trace.csv") 100 60 > $R/gpurun_out/timeline_$s.txt
  python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/line_$s.json')); print('$s', d['ms_per_step'])"
done
