#!/usr/bin/env python3
"""-s 1 (every m-mer selected) over 100 records of 5 Mbp: three scan calls, for a kernel trace; usage: tools/exp/scan_s1_trace.py [total_bp] [s]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
total = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
s = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
R = 5_000_000; n_rec = total // R
off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * R
ctx = sp.Context(0)
p = sp.make_params(31, 11, s)
for _ in range(2): ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), n_rec)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): _, n_out = ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), n_rec)
torch.cuda.synchronize()
print(json.dumps({"s": s, "ms": (time.perf_counter() - t0) * 1e3 / 3, "superkmers": int(n_out)}))
