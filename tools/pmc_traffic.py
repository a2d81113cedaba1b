#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate rocprofv3 --pmc passes of bench.py) -> the JSON that
bench.py's roofline.traffic is looked up from.  gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE (KiB) reports
half of the bytes of a wide coalesced streaming read, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
usage: tools/pmc_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <out.json> [bench command text]"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    by = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("spsp::", "")
        by[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for name, rows in by.items():
        from collections import Counter
        gmax = Counter(g for g, v in rows if v >= 0.5 * max(x for _, x in rows)).most_common(1)[0][0]   # the steps' grid
        vals = sorted(v for g, v in rows if g == gmax)
        vals = [v for v in vals if v >= 0.5 * vals[-1]] or vals
        out[name] = sum(vals) / len(vals)
    return out


def commit():
    """the code state the counters were collected on (set by the caller: the GPU box has no .git)"""
    import os
    return os.environ.get("SPSP_COMMIT")


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("k_"):
            continue
        f, w = fetch.get(name, 0.0), write.get(name, 0.0)
        kernels[name] = {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                         "hbm_bytes_per_launch_corrected": int(round(2 * f * 1024 + w * 1024))}
    doc = {
        "command": sys.argv[4] if len(sys.argv) > 4 else
        "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline",
        "note": "FETCH_SIZE/WRITE_SIZE are reported in KiB. MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly 1/2 "
                "of the bytes of a wide coalesced streaming read (16 B/lane), so fetch_bytes_corrected = 2 * FETCH_SIZE * 1024; "
                "WRITE_SIZE is exact for 16 B/lane stores and per-dword atomics. Per-launch averages over the timed-step "
                "launches (most frequent grid among the large ones, >= half of the largest value).",
        "workload": {"genomes": 100, "genome_len": 5000000, "k": 31, "m": 11, "s": 1000.0, "scan_mode": "default"},
        "commit": commit(),
        "kernels": kernels,
    }
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(kernels.get("k_dense_pair")))


if __name__ == "__main__":
    main()
