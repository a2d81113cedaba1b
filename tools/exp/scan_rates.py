#!/usr/bin/env python3
"""The minimizer scan under other parameters: 100 records of 5 Mbp of random bases, one spsp_scan_device call per
(k, m, s) -- from the sparse default (s = 1000) to every k-mer selected (s = 1).
usage: tools/exp/scan_rates.py [total_bp=500000000]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
R = 5_000_000
n_rec = total // R
off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * R
fresh = os.environ.get("SCAN_RATES_FRESH", "1") == "1"      # a context per configuration (0: one context for all, in order)
ctx = None if fresh else sp.Context(0)
for k, m, s in [(31, 11, 1000), (31, 11, 100), (31, 11, 10), (31, 11, 2), (31, 11, 1), (63, 15, 100), (63, 15, 4), (21, 9, 1000), (21, 9, 5), (31, 15, 50)]:
    if fresh:
        ctx = sp.Context(0)
    p = sp.make_params(k, m, s)
    call = lambda: ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), n_rec)  # noqa: E731
    _, n_out = call()
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    if fresh:
        ctx.close()
    print(json.dumps({"k": k, "m": m, "s": s, "superkmers": int(n_out), "ms": round(ms, 3), "kmers_per_s": round((R - k + 1) * n_rec / ms * 1e3, 0)}), flush=True)
if not fresh:
    ctx.close()
