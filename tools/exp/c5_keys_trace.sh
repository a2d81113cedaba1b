# configs[4]-shape key extraction: timings + kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/tools/exp/c5_keys.py 2>/dev/null | tail -24
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt5k
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5k -o r -- python3 $R/tools/exp/c5_keys.py > /dev/null 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/kt5k -name "*kernel_trace.csv") /tmp/kt5k/s.md > /dev/null; grep "k_" /tmp/kt5k/s.md | cut -c1-120
