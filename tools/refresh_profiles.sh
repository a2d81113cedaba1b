#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: refreshes gpurun_out/prof_<tag>/ with
#   bench    bench line (plain run), kernel-trace summary of the same command, bench line under rocprof,
#            FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), the SQ counters of the dense kernel
#   compare  the same for the comparator at BASELINE configs[2], true shape (tools/c3_compare.py 1000)
#   c4       ... at configs[3] scale on one GPU (tools/c4_compare.py 10000)
#   c5       the configs[4]-shape scan (tools/c5_scan.py 4 4): kernel trace, FETCH/WRITE, SQ counters of k_dense_bloom
#   fetchcal FETCH_SIZE / WRITE_SIZE per byte for known access shapes (tools/exp/exp_fetchcal)
# usage: tools/refresh_profiles.sh <tag> [bench|compare|c4|c5|fetchcal|all]
set -o pipefail
tag=${1:-r04}
what=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SQ_SETS=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_CYCLES")
# pmc_pair <prefix> <command...>: FETCH_SIZE and WRITE_SIZE passes -> $out/<prefix>_FETCH_SIZE.csv / _WRITE_SIZE.csv
pmc_pair() {
  pre=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${pre}_$c && timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${pre}_$c -o r -- "$@" > /dev/null 2>&1 || return 1
    cp $(find /tmp/pmc_${pre}_$c -name "*counter_collection.csv") $out/${pre}_$c.csv
  done
}
# sq_sets <prefix> <command...>: the three SQ counter passes -> /tmp/pmc_<prefix>_sq_*/
sq_sets() {
  pre=$1; shift
  i=0
  for set in "${SQ_SETS[@]}"; do
    d=/tmp/pmc_${pre}_sq_$i; rm -rf $d; i=$((i+1))
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $d -o r -- "$@" > /dev/null 2>&1 || return 1
  done
}
if [ "$what" = all ] || [ "$what" = bench ]; then
timeout -k 10 900 python3 $R/bench.py > $out/bench_line.json 2> $out/bench.err || exit 1
echo "bench line done"
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 $R/bench.py --no-cpu-baseline --no-extras > $out/bench_line_under_rocprof.json 2>/dev/null || exit 1
kt=$(find /tmp/kt -name "*kernel_trace.csv"); ks=$(find /tmp/kt -name "*kernel_stats.csv")
python3 $R/tools/prof_summary.py $kt $out/bench_kernel_summary.md > /dev/null && cp $ks $out/bench_kernel_stats.csv
python3 $R/tools/timeline_mid.py $kt 400 48 > $out/bench_timeline.txt
echo "kernel trace done"
cmd="python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras"
pmc_pair bench $cmd || exit 1
python3 $R/tools/pmc_traffic.py $out/bench_FETCH_SIZE.csv $out/bench_WRITE_SIZE.csv $out/pmc_hbm_traffic.json > /dev/null
echo "traffic done"
sq_sets bench $cmd || exit 1
python3 $R/tools/pmc_sq.py --json $out/pmc_sq_bench.json --command "rocprofv3 --pmc <4 SQ counters per pass> -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras" "k_dense_pair<true, false>" $(find /tmp/pmc_bench_sq_* -name "*counter_collection.csv") > $out/pmc_sq_k_dense_pair.txt
rm -f $out/bench_FETCH_SIZE.csv $out/bench_WRITE_SIZE.csv
echo "sq done"
fi
if [ "$what" = all ] || [ "$what" = compare ]; then
cmd="python3 $R/tools/c3_compare.py 1000 10"
rm -rf /tmp/ktc && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktc -o r -- $cmd > $out/compare_line_under_rocprof.txt 2>/dev/null || exit 1
python3 $R/tools/prof_summary.py $(find /tmp/ktc -name "*kernel_trace.csv") $out/compare_kernel_summary.md > /dev/null
pmc_pair cmp $cmd || exit 1
python3 $R/tools/pmc_traffic.py $out/cmp_FETCH_SIZE.csv $out/cmp_WRITE_SIZE.csv $out/compare_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/c3_compare.py 1000 10" '{"sketches": 1000, "shape": "BASELINE configs[2] at its true shape: 1000 genomes of L ~ U[2, 8] Mbp, 50 families x 20, k=31 m=11 s=1000, sketched on the device"}' > /dev/null
rm -f $out/cmp_FETCH_SIZE.csv $out/cmp_WRITE_SIZE.csv
echo "compare done"
fi
if [ "$what" = all ] || [ "$what" = c4 ]; then
cmd="python3 $R/tools/c4_compare.py 10000 5"
rm -rf /tmp/kt4 && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -o r -- $cmd > $out/c4_line_under_rocprof.json 2>/dev/null || exit 1
python3 $R/tools/prof_summary.py $(find /tmp/kt4 -name "*kernel_trace.csv") $out/c4_kernel_summary.md > /dev/null
pmc_pair c4 $cmd || exit 1
python3 $R/tools/pmc_traffic.py $out/c4_FETCH_SIZE.csv $out/c4_WRITE_SIZE.csv $out/c4_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/c4_compare.py 10000 5" '{"sketches": 10000, "shape": "BASELINE configs[3]: 500 families x 20 synthesised directly, ~5 000 keys per sketch, k=31 m=11"}' > /dev/null
rm -f $out/c4_FETCH_SIZE.csv $out/c4_WRITE_SIZE.csv
echo "c4 done"
fi
if [ "$what" = all ] || [ "$what" = c5 ]; then
cmd="python3 $R/tools/c5_scan.py 4 4"
rm -rf /tmp/kt5 && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -o r -- $cmd > $out/c5_run_under_rocprof.json 2>/dev/null || exit 1
python3 $R/tools/prof_summary.py $(find /tmp/kt5 -name "*kernel_trace.csv") $out/c5_kernel_summary.md > /dev/null
pmc_pair c5 $cmd || exit 1
python3 $R/tools/pmc_traffic.py $out/c5_FETCH_SIZE.csv $out/c5_WRITE_SIZE.csv $out/c5_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/c5_scan.py 4 4" '{"bases_per_launch": 4000000000, "k": 63, "m": 15, "s": 100.0}' > /dev/null
sq_sets c5 $cmd || exit 1
python3 $R/tools/pmc_sq.py --json $out/pmc_sq_c5.json --command "rocprofv3 --pmc <4 SQ counters per pass> -- python3 tools/c5_scan.py 4 4" "k_dense_bloom<15" $(find /tmp/pmc_c5_sq_* -name "*counter_collection.csv") > $out/pmc_sq_k_dense_bloom.txt
rm -f $out/c5_FETCH_SIZE.csv $out/c5_WRITE_SIZE.csv
echo "c5 done"
fi
if [ "$what" = all ] || [ "$what" = fetchcal ]; then
cmd="$R/tools/exp/exp_fetchcal 67108864"
pmc_pair cal $cmd || exit 1
python3 $R/tools/fetchcal_summary.py $out/cal_FETCH_SIZE.csv $out/cal_WRITE_SIZE.csv 67108864 $out/fetch_calibration.md > /dev/null
rm -f $out/cal_FETCH_SIZE.csv $out/cal_WRITE_SIZE.csv
echo "fetchcal done"
fi
ls -la $out
