#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: refreshes gpurun_out/prof_<tag>/ with
#   bench line (plain run), kernel-trace summary of the same command, bench line under rocprof,
#   FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), the SQ counters of the dense kernel,
#   and the same for the comparator at BASELINE configs[2] scale (tests/tools/compare_bench.py 1000).
# usage: tools/refresh_profiles.sh <tag> [bench|compare|all]
set -o pipefail
tag=${1:-r02}
what=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$what" = all ] || [ "$what" = bench ]; then
timeout -k 10 400 python3 $R/bench.py > $out/bench_line.json 2> $out/bench.err || exit 1
echo "bench line done"
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 $R/bench.py --no-cpu-baseline --no-extras > $out/bench_line_under_rocprof.json 2>/dev/null || exit 1
kt=$(find /tmp/kt -name "*kernel_trace.csv"); ks=$(find /tmp/kt -name "*kernel_stats.csv")
python3 $R/tools/prof_summary.py $kt $out/bench_kernel_summary.md > /dev/null && cp $ks $out/bench_kernel_stats.csv
python3 $R/tools/timeline_mid.py $kt 400 48 > $out/bench_timeline.txt
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv") $out/pmc_$c.csv
done
python3 $R/tools/pmc_traffic.py $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv $out/pmc_hbm_traffic.json > /dev/null
echo "traffic done"
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_CYCLES"; do
  d=/tmp/pmc_sq_$(echo $set | cut -d" " -f1); rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
done
python3 $R/tools/pmc_sq.py "k_dense_pair<true, false>" $(find /tmp/pmc_sq_* -name "*counter_collection.csv") > $out/pmc_sq_k_dense_pair.txt
python3 $R/tools/pmc_sq.py "k_dense_pair<true, false>" $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv >> $out/pmc_sq_k_dense_pair.txt
rm -f $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv
echo "sq done"
fi
if [ "$what" = all ] || [ "$what" = compare ]; then
cmd="python3 $R/tests/tools/compare_bench.py 1000 0"
rm -rf /tmp/ktc && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktc -o r -- $cmd > $out/compare_line_under_rocprof.txt 2>/dev/null || exit 1
python3 $R/tools/prof_summary.py $(find /tmp/ktc -name "*kernel_trace.csv") $out/compare_kernel_summary.md > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$c && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcc_$c -o r -- $cmd > /dev/null 2>&1 || exit 1
  cp $(find /tmp/pmcc_$c -name "*counter_collection.csv") $out/cpmc_$c.csv
done
python3 $R/tools/pmc_traffic.py $out/cpmc_FETCH_SIZE.csv $out/cpmc_WRITE_SIZE.csv $out/compare_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tests/tools/compare_bench.py 1000 0" > /dev/null
rm -f $out/cpmc_FETCH_SIZE.csv $out/cpmc_WRITE_SIZE.csv
echo "compare done"
fi
ls -la $out
