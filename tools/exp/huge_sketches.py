#!/usr/bin/env python3
"""A FEW sketches of millions of keys each (large genomes at a fine sampling rate): G genomes of L bases derived from one
ancestor (substitutions at 0.1 % ... 2 %), scanned and turned into comparator keys on the device, compared all-vs-all;
every pair checked against set algebra on the keys (torch).
usage: tools/exp/huge_sketches.py [G=8] [L=300000000] [s=100]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000_000
s = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
anc = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
bases = torch.empty(G * L, dtype=torch.uint8, device=dev)
for i in range(G):
    mu = [0.001, 0.005, 0.02][i % 3]
    hit = torch.rand(L, generator=g, device=dev) < mu
    code = torch.where(hit, (anc + torch.randint(1, 4, (L,), generator=g, device=dev, dtype=torch.uint8)) % 4, anc)
    bases[i * L:(i + 1) * L] = acgt[code.long()]
    del hit, code
del anc
off = torch.arange(0, G + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize()                                  # (the library's stream is not torch's: the bases must be there)
print("bases head %s tail %s, bad bytes %d" % (bytes(bases[:24].cpu().numpy()), bytes(bases[-24:].cpu().numpy()),
                                                 int(sum(int(((bases[i * L:(i + 1) * L] != 65) & (bases[i * L:(i + 1) * L] != 67) & (bases[i * L:(i + 1) * L] != 71) & (bases[i * L:(i + 1) * L] != 84)).sum().item()) for i in range(G)))), flush=True)
ctx = sp.Context(0)
p = sp.make_params(31, 11, s)
t0 = time.perf_counter()
d_sk, n_sk = ctx.scan_device(p, bases.data_ptr(), bases.numel(), off.data_ptr(), G)
t1 = time.perf_counter()
d_mn, d_lo, _, koff = ctx.sketch_keys_device(p, bases.data_ptr(), bases.numel(), off.data_ptr(), d_sk, n_sk, np.arange(G + 1, dtype=np.uint32))
t2 = time.perf_counter()
S = int(koff[-1])
print(json.dumps({"genomes": G, "bases_each": L, "s": s, "superkmers": int(n_sk), "keys": S, "scan_ms": round((t1 - t0) * 1e3, 2), "keys_ms": round((t2 - t1) * 1e3, 2)}), flush=True)
# the keys belong to the context until its next key extraction: copy them out
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
mn = torch.empty(S, dtype=torch.int32, device=dev); lo = torch.empty(S, dtype=torch.int64, device=dev)
hip.hipMemcpy(ctypes.c_void_p(mn.data_ptr()), ctypes.c_void_p(d_mn), ctypes.c_size_t(S * 4), 3)
hip.hipMemcpy(ctypes.c_void_p(lo.data_ptr()), ctypes.c_void_p(d_lo), ctypes.c_size_t(S * 8), 3)
d_inter = torch.zeros((G, G), dtype=torch.int32, device=dev)
ts = []
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.compare_device(31, mn.data_ptr(), lo.data_ptr(), None, koff, G, 0, 1, d_inter.data_ptr())
    ts.append((time.perf_counter() - t0) * 1e3)
got = d_inter.cpu().numpy()
wrong = 0
key = torch.stack([mn.to(torch.int64) & 0xffffffff, lo], 1)
o = koff.astype(np.int64)
for i in range(G):
    for j in range(i + 1, G):
        both = torch.cat([key[o[i]:o[i + 1]], key[o[j]:o[j + 1]]]).unique(dim=0).shape[0]
        want = int(o[i + 1] - o[i] + o[j + 1] - o[j]) - both
        wrong += int(got[i, j]) != want
print(json.dumps({"compare_ms": [round(t, 2) for t in ts], "keys_per_sketch": int(S // G), "pairs_wrong": wrong, "inter_0_1": int(got[0, 1]), "inter_0_2": int(got[0, min(2, G - 1)])}), flush=True)
sys.exit(1 if wrong else 0)
