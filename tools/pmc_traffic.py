#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate rocprofv3 --pmc passes) -> the JSON bench.py's
`roofline.traffic` figures are looked up from.

Correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a WIDE COALESCED
STREAMING read (16 B per lane); other access shapes are uncalibrated there.  tools/exp/exp_fetchcal.hip measures them
on known byte counts (profiles/r03_fetch_calibration.md); the factors below are what it found.  A kernel is classed by
its dominant read shape:
  stream  -- coalesced streams (dense passes, pack, ingest, scatter's key reads): fetch bytes = FETCH_FACTOR_STREAM x FETCH_SIZE
  gather  -- scattered 4-16 byte reads (list references, lists, records by slot): fetch bytes = FETCH_FACTOR_GATHER x FETCH_SIZE
             (the counter shows 64 B per random access; by their time -- 2^26 accesses in 1.31 ms = the 6.5 TB/s read ceiling at
             128 B each -- every one moves a whole 128-byte line: the factor is 2 here as well)
WRITE_SIZE is taken as reported (exact for 16 B-per-lane stores and per-dword atomics; scattered 16-byte stores are
counted as the 32-byte sectors they touch -- that IS the traffic).
usage: tools/pmc_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <out.json> [command text] [workload json]"""
import csv
import json
import os
import re
import sys
from collections import Counter, defaultdict

FETCH_FACTOR_STREAM = 2.0
FETCH_FACTOR_GATHER = float(os.environ.get("SPSP_FETCH_FACTOR_GATHER", "2.0"))   # see profiles/r03_fetch_calibration.md: 64 B counted, 128 B moved (timing)
GATHER_KERNELS = ("k_accumulate_sparse", "k_accumulate", "k_parts_group", "k_fill", "k_fill_sparse", "k_insert", "k_insert_sparse",
                  "k_resolve", "k_compact", "k_decode_emit", "k_abund")


def shape_of(name):
    base = name.split("<")[0]
    return "gather" if base in GATHER_KERNELS else "stream"


def per_kernel(path, counter):
    by = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("spsp::", "")
        by[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for name, rows in by.items():
        gmax = Counter(g for g, v in rows if v >= 0.5 * max(x for _, x in rows)).most_common(1)[0][0]   # the steps' grid
        vals = sorted(v for g, v in rows if g == gmax)
        vals = [v for v in vals if v >= 0.5 * vals[-1]] or vals
        out[name] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("k_"):
            continue
        f, nf = fetch.get(name, (0.0, 0))
        w, _ = write.get(name, (0.0, 0))
        shape = shape_of(name)
        factor = FETCH_FACTOR_GATHER if shape == "gather" else FETCH_FACTOR_STREAM
        kernels[name] = {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "launches_averaged": nf,
                         "read_shape": shape, "fetch_factor": factor,
                         "hbm_bytes_per_launch_corrected": int(round(factor * f * 1024 + w * 1024)),
                         "hbm_bytes_per_launch_if_fetch_x2": int(round(2 * f * 1024 + w * 1024)),
                         "hbm_bytes_per_launch_raw": int(round(f * 1024 + w * 1024))}
    doc = {
        "command": sys.argv[4] if len(sys.argv) > 4 else
        "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras",
        "note": "FETCH_SIZE/WRITE_SIZE are reported in KiB; per-launch averages over the launches with the most frequent grid among "
                "the large ones.  fetch bytes = fetch_factor x FETCH_SIZE: 2.0 for kernels whose reads are coalesced streams "
                "(MI355X_MICROARCH.md), %.1f for kernels whose reads are scattered gathers (profiles/r03_fetch_calibration.md); "
                "both uncorrected and x2 figures are kept beside the chosen one." % FETCH_FACTOR_GATHER,
        "workload": json.loads(sys.argv[5]) if len(sys.argv) > 5 else
        {"genomes": 100, "genome_len": 5000000, "k": 31, "m": 11, "s": 1000.0, "scan_mode": "default"},
        "commit": os.environ.get("SPSP_COMMIT"),     # the code state the counters were collected on (the GPU box has no .git)
        "kernels": kernels,
    }
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    for k in ("k_dense_pair", "k_dense_bloom", "k_accumulate_sparse"):
        for name in kernels:
            if name.startswith(k):
                print(name, json.dumps(kernels[name]))


if __name__ == "__main__":
    main()
