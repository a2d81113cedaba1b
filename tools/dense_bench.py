#!/usr/bin/env python3
"""Micro-benchmark of the dense scan stage only (spsp_scan_hits_device):
prints the HIP-event time of k_dense per launch for each variant.

usage: tools/dense_bench.py [n_bases] [m] [s] [k]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import supersampler_amd as sp  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 11
s = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
k = int(sys.argv[4]) if len(sys.argv) > 4 else 31
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
bases = acgt[torch.randint(0, 4, (n,), device=dev, generator=g)]
torch.cuda.synchronize()
cus = int(os.environ.get("DENSE_CUS", "0"))          # DENSE_CUS=192: on a stream that owns the last 192 CUs, two workgroups each
if cus:
    h = sp.stream_create_cus(0, 256 - cus, cus)
    ctx = sp.Context(0, h)
    ctx.set_cu_count(cus, int(os.environ.get("DENSE_WG_PER_CU", "2")))
else:
    ctx = sp.Context(0, torch.cuda.current_stream().cuda_stream or None)
d_packed = ctx.pack_bases_device(bases.data_ptr(), n)
for name, flag in (("pair", sp.SPSP_SCAN_PAIR_FILTER), ("pair2bit", sp.SPSP_SCAN_PAIR_FILTER | sp.SPSP_SCAN_PACKED_INPUT), ("single", sp.SPSP_SCAN_LDS_FILTER),
                   ("bloom", sp.SPSP_SCAN_BLOOM_FILTER), ("direct", sp.SPSP_SCAN_DIRECT_HASH), ("default", sp.SPSP_SCAN_DEFAULT)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != name:
        continue
    p = sp.make_params(k, m, s, flags=flag)
    src = d_packed if flag & sp.SPSP_SCAN_PACKED_INPUT else bases.data_ptr()
    hits = ctx.scan_hits_device(p, src, n)
    ctx.timing_enable(True); ctx.timing_read()
    reps = 10
    for _ in range(reps):
        ctx.scan_hits_device(p, src, n)
    t = ctx.timing_read()
    ms = t["dense_ms"] / t["dense_launches"]
    byts = n / 4 if flag & sp.SPSP_SCAN_PACKED_INPUT else n
    print("%-9s hits=%d  dense %.4f ms  -> %.2e positions/s, %.1f GB/s (%.1f%% of 8 TB/s)" % (name, hits, ms, n / ms * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 80.0), flush=True)
