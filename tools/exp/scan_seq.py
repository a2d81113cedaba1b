#!/usr/bin/env python3
"""a sequence of scans on ONE context, per-call wall time.  usage: scan_seq.py total_bp k,m,s,reps [k,m,s,reps ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

total = int(sys.argv[1])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
R = 5_000_000
n_rec = total // R
off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * R
ctx = sp.Context(0)
for spec in sys.argv[2:]:
    k, m, s, reps = spec.split(",")
    p = sp.make_params(int(k), int(m), float(s))
    ts = []
    for r in range(int(reps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, n_out = ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), n_rec)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(spec, n_out, " ".join("%.3f" % t for t in ts), flush=True)
ctx.close()
