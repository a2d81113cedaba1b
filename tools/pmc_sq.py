#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (SQ counters) per kernel: per-launch averages over the launches
with the most frequent grid (the steps).  Text to stdout; with --json <out.json> also the file bench.py's
`valu_issue_frac` is looked up from.
usage: tools/pmc_sq.py [--json out.json] [--command "text"] <kernel substring>[;<kernel substring>...] <counter_collection.csv> [more csv ...]"""
import csv
import json
import os
import re
import sys
from collections import Counter, defaultdict


def collect(want, paths):
    acc = defaultdict(list)
    for path in paths:
        rows = [r for r in csv.DictReader(open(path)) if want in r["Kernel_Name"]]
        if not rows:
            continue
        gmax = Counter(int(r["Grid_Size"]) for r in rows).most_common(1)[0][0]    # the steps' grid (extras launch others)
        per = defaultdict(dict)
        for r in rows:
            if int(r["Grid_Size"]) == gmax:
                per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        for d in per.values():
            for k, v in d.items():
                acc[k].append(v)
    out = {}
    for k in sorted(acc):
        v = sorted(acc[k])
        v = [x for x in v if x >= 0.5 * v[-1]]   # timed-step launches
        out[k] = (sum(v) / len(v), len(v))
    return out


def main():
    args = sys.argv[1:]
    out_json = command = None
    while args and args[0].startswith("--"):
        if args[0] == "--json":
            out_json = args[1]
        elif args[0] == "--command":
            command = args[1]
        args = args[2:]
    kernels = {}
    for want in args[0].split(";"):
        c = collect(want, args[1:])
        if not c:
            continue
        name = re.sub(r"\(.*", "", want)
        print("== %s" % want)
        for k, (v, n) in c.items():
            print("%-28s %16.1f  (n=%d)" % (k, v, n))
        kernels[name] = {k: v for k, (v, n) in c.items()}
        kernels[name]["launches_averaged"] = min(n for _, n in c.values())
        if "SQ_INSTS_VALU" in kernels[name] and "SQ_BUSY_CU_CYCLES" in kernels[name]:
            f = kernels[name]["SQ_INSTS_VALU"] / (4.0 * kernels[name]["SQ_BUSY_CU_CYCLES"])
            print("%-28s %16.3f  (4-cycle issue; %.3f with 2-cycle issue)" % ("valu_issue_frac", 4 * f, 2 * f))
    if out_json:
        json.dump({"command": command, "commit": os.environ.get("SPSP_COMMIT"),
                   "note": "per-launch averages of SQ counters (rocprofv3 --pmc, several passes of the same command, 4 counters each); "
                           "SQ_INSTS_* are wave-instructions, SQ_BUSY_CU_CYCLES cycles summed over busy CUs, SQ_WAVE_CYCLES / SQ_WAIT_* / "
                           "SQ_ACTIVE_INST_* quad-cycles (MI355X_MICROARCH.md)", "kernels": kernels}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
