#!/usr/bin/env python3
"""Scan + comparator keys on the device (spsp_scan_device + spsp_sketch_keys_device) under other sampling rates:
100 genomes of 5 Mbp (one record each) of random bases, k=31 m=11, s from 1000 down to 1, sorted and unordered keys.
usage: tools/exp/keys_rates.py [genomes=100] [s,s,...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rates = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1000, 100, 10, 2, 1]
R = 5_000_000
total = G * R
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), generator=g, device=dev)]
off = torch.arange(0, G + 1, dtype=torch.int64, device=dev) * R
first_rec = np.arange(G + 1, dtype=np.uint32)
ctx = sp.Context(0)
for s in rates:
    p = sp.make_params(31, 11, s)
    for unordered in (False, True):
        ts, tk = [], []
        for r in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_sk, n_sk = ctx.scan_device(p, bases.data_ptr(), total, off.data_ptr(), G)
            t1 = time.perf_counter()
            _, _, _, koff = ctx.sketch_keys_device(p, bases.data_ptr(), total, off.data_ptr(), d_sk, n_sk, first_rec, unordered=unordered)
            t2 = time.perf_counter()
            ts.append((t1 - t0) * 1e3); tk.append((t2 - t1) * 1e3)
        print(json.dumps({"s": s, "unordered": unordered, "superkmers": int(n_sk), "keys": int(koff[-1]), "big_genomes": ctx.sketch_keys_big_genomes(),
                          "scan_ms": round(min(ts[1:]), 3), "keys_ms": round(min(tk[1:]), 3), "keys_first_ms": round(tk[0], 3),
                          "keys_per_s": round(int(koff[-1]) / min(tk[1:]) * 1e3, 0)}), flush=True)
ctx.close()
