#!/usr/bin/env python3
"""Print the GPU timeline of the last few steps from a rocprofv3 --kernel-trace CSV: start (us, relative), duration,
queue and kernel, so gaps and overlaps between streams are visible.
usage: tools/timeline.py <kernel_trace.csv> [n_dense_steps_from_end=3]"""
import csv
import re
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n_back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    # dense passes of whole steps (followed by the sparse stages), not the dense-only launches of bench.py's "alone" loop
    dense = [i for i, r in enumerate(rows) if "k_dense" in r["Kernel_Name"] and i + 1 < len(rows) and
             any("k_compact" in x["Kernel_Name"] or "k_expand" in x["Kernel_Name"] for x in rows[i + 1:i + 4])]
    first = dense[-n_back] if len(dense) >= n_back else 0
    t0 = int(rows[first]["Start_Timestamp"])
    shown = 0
    for r in rows[first:]:
        if "k_sum_counts" in r["Kernel_Name"] or shown > 60:
            break
        shown += 1
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("spsp::", "")[:40]
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%9.1f %8.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))


if __name__ == "__main__":
    main()
