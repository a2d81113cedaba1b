#!/bin/bash
# kernel-trace timeline of the bench step for a given small-CU count / library.  usage: bash tools/trace_small.sh <cus> [lib.so]
R=$PWD; export BENCH_SMALL_CUS=$1; [ -n "$2" ] && export SPSP_LIB=$R/$2
tag=$1_$(basename ${2:-new} .so)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o r -- python3 $R/bench.py --experiment --steps 100 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/kt_$tag -name "*kernel_trace.csv") > $R/gpurun_out/summary_$tag.md
grep -E "k_parts|k_dense_pair<true, false> \|" $R/gpurun_out/summary_$tag.md | cut -c1-110
