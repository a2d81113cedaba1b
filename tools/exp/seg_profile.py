#!/usr/bin/env python3
"""the segment kernels under rocprofv3: the scan at -s 1 over 500 Mbp (k_seg_scan<false>, <true>) and the statistics pass over 30 Mbp
(k_seg_count, k_seg_tail); usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/exp/seg_profile.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
total, R = 500_000_000, 5_000_000
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
off = torch.arange(0, total // R + 1, dtype=torch.int64, device=dev) * R
ctx = sp.Context(0)
p = sp.make_params(31, 11, 1.0)
for _ in range(5):
    _, n = ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), total // R)
for _ in range(5):
    c = ctx.count_superkmers_device(p, bases.data_ptr(), 30_000_000, off.data_ptr(), 6)
print("super-k-mers at -s 1:", n, "; all super-k-mers of 30 Mbp:", c)
