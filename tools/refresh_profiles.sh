#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: refreshes gpurun_out/prof_<tag>/ with
#   bench line (plain run), kernel-trace summary of the same command, bench line under rocprof,
#   FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), and the SQ counters of the dense kernel.
# usage: tools/refresh_profiles.sh <tag>
set -o pipefail
tag=${1:-r01_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $out/bench_line.json 2> $out/bench.err || exit 1
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 $R/bench.py --no-cpu-baseline > $out/bench_line_under_rocprof.json 2>/dev/null || exit 1
kt=$(find /tmp/kt -name "*kernel_trace.csv"); ks=$(find /tmp/kt -name "*kernel_stats.csv")
python3 $R/tools/prof_summary.py $kt $out/bench_kernel_summary.md > /dev/null && cp $ks $out/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv") $out/pmc_$c.csv
done
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_CYCLES"; do
  d=/tmp/pmc_sq_$(echo $set | cut -d" " -f1); rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
done
python3 $R/tools/pmc_sq.py k_dense_pair $(find /tmp/pmc_sq_* -name "*counter_collection.csv") > $out/pmc_sq_k_dense_pair.txt
python3 $R/tools/pmc_sq.py k_dense_pair $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv >> $out/pmc_sq_k_dense_pair.txt
ls -la $out
