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
    import time
    d = torch.full((n, n), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    call = lambda: ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, first, stride, d.data_ptr(), n_query=limit)  # noqa: E731
    for _ in range(3):
        call()
    # (1) host wall clock per call, no event brackets on the stream: what a rank's step takes; (2) the same calls with the
    # kernel brackets: where the time goes (the brackets themselves stretch the pipeline: never mix the two kinds of figure)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / reps
    ctx.timing_enable(True, sp.TIME_ALL)
    ctx.timing_read()
    for _ in range(reps):
        call()
    t = ctx.timing_read()
    ctx.timing_enable(False)
    torch.cuda.synchronize()
    per = lambda k: t[k + "_ms"] / max(1, t[k + "_launches"])  # noqa: E731
    return d.cpu().numpy(), {"wall_ms": round(wall_ms, 4), "pipeline_ms": round(t["compare_ms"] / reps, 4), "scatter_ms": round(per("scatter"), 4),
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
        t["share_of_full"] = round(t["pipeline_ms"] / t_full["pipeline_ms"], 3)                  # bracketed pipeline against bracketed pipeline
        t["wall_share_of_full"] = round(t["wall_ms"] / t_full["wall_ms"], 3)                    # host wall clock against host wall clock
        out["%s_rank%d" % (form, r)] = t
# ---- key-partitioned split
import time  # noqa: E402
mn_all, lo_all = D.minimizer, D.kmer_lo
off = D.sk_off.astype(np.int64)
blocks = []
for sdr in range(G):
    a, b = int(off[sdr * per]), int(off[(sdr + 1) * per])
    blocks.append((mn_all[a:b].contiguous(), lo_all[a:b].contiguous(), (off[sdr * per:(sdr + 1) * per + 1] - a).astype(np.uint64)))
most = max(int(x[2][-1]) for x in blocks)
cap = int(most / G * 1.25) + 4096
slot_sz = sp.slot_bytes(per, cap, 31)
sends = [torch.zeros(G * slot_sz, dtype=torch.uint8, device=dev) for _ in range(G)]
part_ms = []
for sdr in range(G):
    bm, bl, bo = blocks[sdr]
    ctx.partition_keys_device(31, bm.data_ptr(), bl.data_ptr(), None, bo, per, G, cap, sends[sdr].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.partition_keys_device(31, bm.data_ptr(), bl.data_ptr(), None, bo, per, G, cap, sends[sdr].data_ptr())
    torch.cuda.synchronize()
    part_ms.append((time.perf_counter() - t0) * 1e3 / reps)
total = torch.zeros((n, n), dtype=torch.int32, device=dev)
d_part = torch.zeros((n, n), dtype=torch.int32, device=dev)
cells = torch.zeros(max(1 << 20, 64 * n), dtype=torch.int64, device=dev)
keyed = {}
for r in range(G):
    recv = torch.cat([sends[sdr][r * slot_sz:(r + 1) * slot_sz] for sdr in range(G)])
    torch.cuda.synchronize()
    for _ in range(2):
        cnt = ctx.compare_slots_cells_device(31, recv.data_ptr(), G, per, cap, d_part.data_ptr(), cells.data_ptr(), cells.numel())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):                           # (1) wall clock, no brackets
        cnt = ctx.compare_slots_cells_device(31, recv.data_ptr(), G, per, cap, d_part.data_ptr(), cells.data_ptr(), cells.numel())
    torch.cuda.synchronize()
    cmp_ms = (time.perf_counter() - t0) * 1e3 / reps
    ctx.timing_enable(True, sp.TIME_ALL)
    ctx.timing_read()
    for _ in range(reps):                           # (2) with the kernel brackets
        cnt = ctx.compare_slots_cells_device(31, recv.data_ptr(), G, per, cap, d_part.data_ptr(), cells.data_ptr(), cells.numel())
    cells_ms = 0.0                                  # (the cells leave the row sums: no pass of their own)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    ctx.matrix_add_cells_device(total.data_ptr(), n, cells.data_ptr(), cnt)
    pk = lambda k: t[k + "_ms"] / max(1, t[k + "_launches"])  # noqa: E731
    keyed["rank%d" % r] = {"partition_own_keys_ms": round(part_ms[r], 4), "compare_slots_cells_wall_ms": round(cmp_ms, 4),
                           "compare_pipeline_ms": round(t["compare_ms"] / max(1, t["compare_calls"]), 4), "scatter_ms": round(pk("scatter"), 4),
                           "group_ms": round(pk("group"), 4), "accumulate_ms": round(pk("accumulate"), 4), "cells": int(cnt),
                           "wall_share_of_full": round((part_ms[r] + cmp_ms + cells_ms) / t_full["wall_ms"], 3),                       # wall against wall
                           "share_of_full": round((t["compare_ms"] / max(1, t["compare_calls"])) / t_full["pipeline_ms"], 3)}   # bracketed against bracketed
torch.cuda.synchronize()
up = np.triu(np.ones((n, n), bool), 1)
keyed["sum_of_partials_equals_full"] = bool((total.cpu().numpy()[up] == full[up]).all())
keyed["slot_cap"], keyed["slot_bytes"] = cap, slot_sz
keyed["largest_share"] = max(v["share_of_full"] for k2, v in keyed.items() if k2.startswith("rank"))
keyed["largest_wall_share"] = max(v["wall_share_of_full"] for k2, v in keyed.items() if k2.startswith("rank"))
out["key_partitioned"] = keyed
print(json.dumps(out))
ctx.close()
