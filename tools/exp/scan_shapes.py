#!/usr/bin/env python3
"""The minimizer scan under other record shapes: 500 Mbp of random bases (100 genomes' worth) cut into records of R
bases each (finished genomes ... contigs of a draft assembly ... reads), one spsp_scan_device call, k=31 m=11 s=1000.
usage: tools/exp/scan_shapes.py [total_bp=500000000] [R,R,...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
Rs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5_000_000, 100_000, 10_000, 1_000, 300, 150, 40]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
ctx = sp.Context(0)
p = sp.make_params(31, 11, 1000)
for R in Rs:
    n_rec = (total + R - 1) // R
    off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * R
    off[-1] = total
    call = lambda: ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), n_rec)  # noqa: E731
    _, n_out = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    kmers = max(0, R - 30) * (total // R) + max(0, total % R - 30)
    print(json.dumps({"record_bp": R, "records": n_rec, "superkmers": int(n_out), "ms": round(ms, 3), "bases_per_s": round(total / ms * 1e3, 0),
                      "kmers_per_s": round(kmers / ms * 1e3, 0)}), flush=True)
ctx.close()
