R=${GRAFT_REPO_ROOT:-$(pwd)}
SPSP_DEBUG_DECODE_TIMES=1 python3 $R/tools/exp/c4_files.py 1 2>&1 | grep -v amdgpu | grep "decode\]\|wall_s\|read_gunzip\|decode_compare\|format_both\|gzip_and"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktf -o r -- python3 $R/tools/exp/c4_files.py 1 > /dev/null 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/ktf -name "*kernel_trace.csv") /tmp/ktf/s.md > /dev/null; grep "k_decode\|k_exclusive\|k_scan_b" /tmp/ktf/s.md | cut -c1-120
