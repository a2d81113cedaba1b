# kernel trace of the closed step with two unordered dense streams (BENCH_DENSE_STREAMS=2): timeline of a window in the middle
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BENCH_DENSE_STREAMS=${1:-2}
rm -rf /tmp/ktd && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktd -o r -- python3 $R/bench.py --experiment --steps 60 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/tds_line.json 2>/dev/null || exit 1
kt=$(find /tmp/ktd -name "*kernel_trace.csv")
python3 $R/tools/timeline_mid.py $kt 320 90 > $R/gpurun_out/tds_timeline_$BENCH_DENSE_STREAMS.txt
tail -c 300 $R/gpurun_out/tds_line.json
