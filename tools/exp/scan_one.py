#!/usr/bin/env python3
"""one (k, m, s) scan of random bases, for rocprofv3.  usage: scan_one.py k m s total_bp [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

k, m, s, total = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
R = int(os.environ.get("SCAN_ONE_RECORD", "5000000"))
n_rec = total // R
off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * R
ctx = sp.Context(0)
p = sp.make_params(k, m, s)
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, n_out = ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), n_rec)
    print("call %d: %d super-k-mers in %.1f ms" % (r, n_out, (time.perf_counter() - t0) * 1e3), flush=True)
ctx.close()
