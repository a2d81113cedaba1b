#!/usr/bin/env python3
"""The comparison at the configs[3] size under other collection shapes: n sketches in families of F for several F
(1 = unrelated genomes ... n = one species, every key held by almost every sketch).  Sampled cells are checked against
set algebra on the keys (torch).
usage: tools/exp/c4_shapes.py [N=10000] [reps=3] [F,F,...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fams = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 20, 50, 100, 200, 500, 2000, n]
use_hi = os.environ.get("SHAPES_HI", "0") == "1"              # k = 63: a second key word (a function of the first, so equal keys stay equal)
n_query = int(os.environ.get("SHAPES_QUERY", "0")) or None      # query mode: rows of the first sketches only
dev = torch.device("cuda", 0)
ctx = sp.Context(0)
if use_hi:
    ctx.compare_keys_unordered(True)                            # (the made-up second word is not in key order; the keys are distinct)
d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
rng = np.random.default_rng(1)
for F in fams:
    D = synth.direct_family_sketches(n, fam_size=F, seed=4, device=dev, skm_range=(120, 360), mus=(0.001, 0.01, 0.05))
    if os.environ.get("SHAPES_SHUFFLE", "0") == "1":            # the sketches in a random order (families no longer side by side)
        perm = rng.permutation(n)
        off0 = D.sk_off.astype(np.int64)
        cnt = np.diff(off0)[perm]
        new_off = np.zeros(n + 1, np.int64)
        new_off[1:] = np.cumsum(cnt)
        src = torch.from_numpy(np.repeat(off0[:-1][perm] - new_off[:-1], cnt)).to(dev) + torch.arange(int(new_off[-1]), device=dev)
        D.minimizer = D.minimizer[src].contiguous()
        D.kmer_lo = D.kmer_lo[src].contiguous()
        D.sk_off = new_off.astype(np.uint64)
        fam_of = (perm // F)
        del src
    else:
        fam_of = np.arange(n) // F
    torch.cuda.synchronize()
    d_hi = ((D.kmer_lo.view(torch.int64) * 0x9E3779B97F4A7C15) >> 3) & 0x0fffffffffffffff if use_hi else None
    call = lambda: ctx.compare_device(63 if use_hi else 31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), d_hi.data_ptr() if use_hi else None, D.sk_off, n, 0, 1,  # noqa: E731
                                      d_inter.data_ptr(), n_query=n_query)
    d_inter.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call()
    first = (time.perf_counter() - t0) * 1e3
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    wall = (time.perf_counter() - t0) * 1e3 / reps
    ctx.timing_enable(True, sp.TIME_ALL)
    ctx.timing_read()
    for _ in range(reps):
        call()
    t = ctx.timing_read()
    ctx.timing_enable(False, 0)
    ms = t["compare_ms"] / reps
    # sampled pairs (inside families and across) against set algebra: keys as (minimizer << 62-bit k-mer) pairs
    key = torch.stack([D.minimizer.to(torch.int64), D.kmer_lo.view(torch.int64) if D.kmer_lo.dtype != torch.int64 else D.kmer_lo], 1)
    off = D.sk_off.astype(np.int64)
    wrong = 0
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, 30), rng.integers(0, n, 30))]
    for a in rng.integers(0, n, 30):                             # pairs inside a family
        mates = np.nonzero(fam_of == fam_of[int(a)])[0]
        pairs.append((int(a), int(mates[int(rng.integers(0, len(mates)))])))
    for a, b in pairs:
        if a == b:
            continue
        i, j = min(a, b), max(a, b)
        if n_query is not None and i >= n_query:
            continue
        ka, kb = key[off[i]:off[i + 1]], key[off[j]:off[j + 1]]
        both = torch.cat([ka, kb]).unique(dim=0).shape[0]
        want = ka.shape[0] + kb.shape[0] - both
        wrong += int(d_inter[i, j].item()) != want
    print(json.dumps({"n": n, "fam_size": F, "k": 63 if use_hi else 31, "n_query": n_query, "keys": int(D.sk_off[-1]), "first_call_ms": round(first, 3), "wall_ms_per_call": round(wall, 3), "pipeline_ms": round(ms, 4),
                      "scatter_ms": round(t["scatter_ms"] / max(1, t["scatter_launches"]), 4), "group_ms": round(t["group_ms"] / max(1, t["group_launches"]), 4),
                      "accumulate_ms": round(t["accumulate_ms"] / max(1, t["accumulate_launches"]), 4),
                      "launches_per_call": {k: v / reps for k, v in t.items() if k.endswith("_launches") and v},
                      "nonzero_pairs": int(torch.count_nonzero(torch.triu(d_inter, 1)).item()), "sampled_pairs_wrong": wrong}), flush=True)
    del D, key
ctx.close()
