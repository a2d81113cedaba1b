#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE CSVs of tools/exp/exp_fetchcal (separate --pmc passes) -> markdown: counter bytes per byte moved.
usage: tools/fetchcal_summary.py <FETCH_SIZE.csv> <WRITE_SIZE.csv> <n_access> <out.md>"""
import csv
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    by = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            by[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in by.items()}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    n = int(sys.argv[3])
    moved = {"k_stream<unsigned int>": ("read", 4), "k_stream<unsigned long>": ("read", 8), "k_stream<HIP_vector_type<unsigned int, 4u> >": ("read", 16),
             "k_gather<unsigned int>": ("read", 4), "k_gather<unsigned long>": ("read", 8), "k_gather<HIP_vector_type<unsigned int, 4u> >": ("read", 16),
             "k_scatter16": ("write", 16), "k_stream_store16": ("write", 16)}
    lines = ["# FETCH_SIZE / WRITE_SIZE per byte moved, by access shape (tools/exp/exp_fetchcal.hip, %d accesses per kernel, 4 GiB buffer)" % n, "",
             "| kernel | bytes moved | FETCH_SIZE bytes | WRITE_SIZE bytes | counter / moved | bytes per access |", "|---|---|---|---|---|---|"]
    for name in sorted(set(f) | set(w)):
        kind, width = None, None
        for key, (kd, wd) in moved.items():
            if name.startswith(key.split("<")[0]) and (("<" not in key) or key.split("<")[1].split(">")[0].split(",")[0] in name):
                if key == name or kind is None:
                    kind, width = kd, wd
        if name in moved:
            kind, width = moved[name]
        if kind is None:
            continue
        mv = n * width
        c = f.get(name, 0.0) if kind == "read" else w.get(name, 0.0)
        lines.append("| %s | %d | %.0f | %.0f | %.3f | %.1f |" % (name, mv, f.get(name, 0.0), w.get(name, 0.0), c / mv, c / n))
    open(sys.argv[4], "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
