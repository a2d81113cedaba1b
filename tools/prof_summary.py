#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV for bench.py runs.

bench.py's setup phase launches the same kernels on single genomes, and its
extras launch the dense pass with other grids (whole-chip context); the steps
are the launches with the MOST FREQUENT grid among the long ones, so the
"steady" columns are restricted to those (that grid and >= half the longest
duration) next to the all-launch totals.  Other grids with at least ten long
launches get a row of their own ("name @grid").

usage: tools/prof_summary.py <kernel_trace.csv> [out.md]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("spsp::", "")
    return name[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    per = defaultdict(list)
    for r in rows:
        if r["Kind"] != "KERNEL_DISPATCH":
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        per[short(r["Kernel_Name"])].append((dur, grid, r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                                            r["Workgroup_Size_X"]))
    lines = ["| kernel | launches | total us | steady launches | steady avg us | steady min us | grid | wg | VGPR | SGPR | static LDS |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    order = sorted(per.items(), key=lambda kv: -sum(d[0] for d in kv[1]))
    for name, ds in order:
        dmax = max(d[0] for d in ds)
        long_ones = [d for d in ds if d[0] * 2 >= dmax]
        by_grid = defaultdict(list)
        for d in long_ones:
            by_grid[d[1]].append(d)
        grids = sorted(by_grid.items(), key=lambda kv: -len(kv[1]))
        tot = sum(d[0] for d in ds) / 1e3
        for gi, (g, steady) in enumerate(grids):
            if gi > 0 and len(steady) < 10:
                continue
            avg = sum(d[0] for d in steady) / len(steady) / 1e3
            mn = min(d[0] for d in steady) / 1e3
            lines.append("| %s | %d | %.1f | %d | %.2f | %.2f | %d | %s | %s | %s | %s |" % (
                name if gi == 0 else "%s @%d" % (name, g), len(ds), tot, len(steady), avg, mn, g, steady[0][5], steady[0][2], steady[0][3], steady[0][4]))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
