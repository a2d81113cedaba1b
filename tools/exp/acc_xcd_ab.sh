# row sums: rows in launch order vs rows of a family behind one XCD's L2; times + TCC hit / miss + FETCH_SIZE of k_accumulate_sparse
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  echo "== SPSP_DEBUG_ACC_XCD=$mode"
  for n in 10000; do SPSP_DEBUG_ACC_XCD=$mode python3 $R/tools/c4_compare.py $n 10; done
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" ; do
    tag=$(echo $c | tr ' ' '_')
    rm -rf /tmp/pmc_acc_$mode_$tag
    SPSP_DEBUG_ACC_XCD=$mode timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_acc_${mode}_$tag -o r -- python3 $R/tools/c4_compare.py 10000 3 > /dev/null 2>&1
    python3 - "$(find /tmp/pmc_acc_${mode}_$tag -name '*counter_collection.csv')" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("spsp::", "")
    if name.startswith(("k_accumulate_sparse", "k_parts_scatter", "k_parts_group")):
        acc[name[:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  ", k, {c: sum(x[-3:]) / len(x[-3:]) for c, x in v.items()})
PY
  done
done
