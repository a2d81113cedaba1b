#!/usr/bin/env python3
"""what a run without a reset costs the scan by segments at -s 1: 100 Mbp of random bases with and without a homopolymer of N bases in the middle"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
total = 100_000_000
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
off = torch.tensor([0, total], dtype=torch.int64, device=dev)
ctx = sp.Context(0)
p = sp.make_params(31, 11, 1.0)
for run in (0, 2000, 8000, 16000, 20000):
    b = bases.clone()
    if run: b[50_000_000:50_000_000 + run] = ord("A")
    torch.cuda.synchronize()
    for _ in range(2): ctx.scan_device(p, b.data_ptr(), total, off.data_ptr(), 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): _, n = ctx.scan_device(p, b.data_ptr(), total, off.data_ptr(), 1)
    torch.cuda.synchronize()
    print("homopolymer of %6d bases: %.2f ms per call, %d super-k-mers" % (run, (time.perf_counter() - t0) / 3 * 1e3, n))
