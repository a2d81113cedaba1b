#!/usr/bin/env python3
"""One rank's share of the all-vs-all comparison after the key all-gather, on ONE GPU, whole chip:
N sketches synthesised directly (families of 20, as bench.py's comparator legs), the full comparison (all rows), then
the calls ranks 0, G/2 and G-1 of G would make -- rows in BLOCKS (a rank's own sketches; the other ranks' keys pass the
Bloom filter) and STRIDED (i % G == rank) -- each checked against the full matrix and timed with the kernel brackets.
usage: tools/rank_share.py [N=800] [G=8] [reps=20]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
D = synth.direct_family_sketches(n, fam_size=20, seed=4, device=dev)
S = int(D.sk_off[-1])
ctx = sp.Context(0)
ctx.compare_keys_unordered(False)


def run(first, stride, limit):
    d = torch.full((n, n), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    call = lambda: ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, first, stride, d.data_ptr(), n_query=limit)  # noqa: E731
    for _ in range(3):
        call()
    ctx.timing_enable(True, sp.TIME_ALL)
    ctx.timing_read()
    for _ in range(reps):
        call()
    t = ctx.timing_read()
    ctx.timing_enable(False)
    torch.cuda.synchronize()
    per = lambda k: t[k + "_ms"] / max(1, t[k + "_launches"])  # noqa: E731
    return d.cpu().numpy(), {"pipeline_ms": round(t["compare_ms"] / reps, 4), "scatter_ms": round(per("scatter"), 4),
                             "group_ms": round(per("group"), 4), "accumulate_ms": round(per("accumulate"), 4)}


full, t_full = run(0, 1, n)
out = {"n": n, "keys": S, "world": G, "full": t_full}
per = n // G
for form in ("block", "strided"):
    for r in sorted({0, G // 2, G - 1}):
        first, stride, limit = (r * per, 1, (r + 1) * per) if form == "block" else (r, G, n)
        got, t = run(first, stride, limit)
        rows = np.arange(first, limit, stride)
        own = np.zeros((n, n), bool)
        own[rows] = np.triu(np.ones((n, n), bool), 1)[rows]
        t["equal_to_full"] = bool((got[own] == full[own]).all() and (got[~own] == -1).all())
        t["share_of_full"] = round(t["pipeline_ms"] / t_full["pipeline_ms"], 3)
        out["%s_rank%d" % (form, r)] = t
print(json.dumps(out))
ctx.close()
