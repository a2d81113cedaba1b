# scatter over consecutive entries (SPSP_DEBUG_TILES=0) vs over tiles of 32 sketches x a key range (default):
# per-kernel times (kernel trace) + TCC hit / miss + FETCH_SIZE + WRITE_SIZE of the comparison's kernels at configs[3]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-10000}
cd /tmp && export TMPDIR=/tmp
for mode in 0 on; do
  if [ $mode = 0 ]; then export SPSP_DEBUG_TILES=0; else unset SPSP_DEBUG_TILES; fi
  echo "== SPSP_DEBUG_TILES=$mode"
  python3 $R/tools/c4_compare.py $N 10
  rm -rf /tmp/kt_t$mode && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_t$mode -o r -- python3 $R/tools/c4_compare.py $N 5 > /dev/null 2>&1
  python3 $R/tools/prof_summary.py $(find /tmp/kt_t$mode -name "*kernel_trace.csv") /tmp/kt_t$mode/summary.md > /dev/null; grep "k_" /tmp/kt_t$mode/summary.md
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $c | tr ' ' '_')
    rm -rf /tmp/pmc_t${mode}_$tag
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_t${mode}_$tag -o r -- python3 $R/tools/c4_compare.py $N 3 > /dev/null 2>&1
    python3 - "$(find /tmp/pmc_t${mode}_$tag -name '*counter_collection.csv')" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spsp::", "")
    if name.startswith(("k_accumulate_sparse", "k_parts_scatter", "k_parts_group", "k_tile")):
        acc[name[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  ", k, {c: sum(x[-3:]) / len(x[-3:]) for c, x in v.items()})
PY
  done
done
