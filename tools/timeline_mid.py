#!/usr/bin/env python3
"""GPU timeline of a few steps from the MIDDLE of a bench.py run (rocprofv3 --kernel-trace CSV): start (us, relative),
duration, queue, kernel -- every dispatch, so gaps and overlaps between streams are visible.
usage: tools/timeline_mid.py <kernel_trace.csv> [first dense launch=100] [rows=70]"""
import csv
import re
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    nth = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 70
    dense = [i for i, r in enumerate(rows) if "k_dense" in r["Kernel_Name"]]
    first = dense[min(nth, len(dense) - 1)]
    t0 = int(rows[first]["Start_Timestamp"])
    for r in rows[first:first + n_rows]:
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("spsp::", "")[:40]
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%9.1f %8.1f %9.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, e / 1e3, r.get("Queue_Id", "?"), name))


if __name__ == "__main__":
    main()
