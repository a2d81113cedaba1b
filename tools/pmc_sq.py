#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for one kernel: per-launch averages of the launches with the most frequent grid (the steps).
usage: tools/pmc_sq.py <kernel substring> <counter_collection.csv> [more csv ...]"""
import csv
import sys
from collections import defaultdict


def main():
    want = sys.argv[1]
    acc = defaultdict(list)
    for path in sys.argv[2:]:
        rows = [r for r in csv.DictReader(open(path)) if want in r["Kernel_Name"]]
        if not rows:
            continue
        from collections import Counter
        gmax = Counter(int(r["Grid_Size"]) for r in rows).most_common(1)[0][0]    # the steps' grid (extras launch others)
        per = defaultdict(dict)
        for r in rows:
            if int(r["Grid_Size"]) == gmax:
                per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        for d in per.values():
            for k, v in d.items():
                acc[k].append(v)
    for k in sorted(acc):
        v = sorted(acc[k])
        v = [x for x in v if x >= 0.5 * v[-1]]   # timed-step launches
        print("%-28s %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
