# kernel trace of a --steps 20 run: the whole timed region (dense launches 305..324 of the process), to see its head and tail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kts && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o r -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/tsr_line.json 2>/dev/null || exit 1
kt=$(find /tmp/kts -name "*kernel_trace.csv")
python3 $R/tools/timeline_mid.py $kt 303 420 > $R/gpurun_out/tsr_timeline.txt
tail -c 200 $R/gpurun_out/tsr_line.json
