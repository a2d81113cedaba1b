# configs[4]-shape scan: kernel time + SQ counters of k_dense_bloom (VALU per position)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/c5_scan.py 4 4 | tail -1
rm -rf /tmp/sq5 && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/sq5 -o r -- python3 $R/tools/c5_scan.py 4 4 > /dev/null 2>&1
python3 $R/tools/pmc_sq.py "k_dense_bloom<15>" $(find /tmp/sq5 -name "*counter_collection.csv")
