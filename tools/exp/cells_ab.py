#!/usr/bin/env python3
"""the all-vs-all comparison of N configs[3]-shaped sketches returned as CELLS (what spsp_compare_files takes from 1 024 files on),
timed by host wall clock and kernel brackets; run under SPSP_DEBUG_ACC_TOUCH=0 / l / s for the A/B.  usage: cells_ab.py [N=10000] [fam=20]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
from supersampler_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
D = synth.direct_family_sketches(n, fam_size=fam, seed=4, device=dev, skm_range=(120, 360))
ctx = sp.Context(0)
scratch = torch.zeros((n, n), dtype=torch.int32, device=dev)   # (the spill route writes the dense matrix: it must be there)
cells = torch.zeros(max(1 << 22, n * fam * 2), dtype=torch.int64, device=dev)
call = lambda: ctx.compare_cells_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())
for _ in range(3): cnt = call()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): call()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 100
ctx.timing_enable(True, sp.TIME_ALL); ctx.timing_read()
for _ in range(10): call()
t = ctx.timing_read(); ctx.timing_enable(False)
c = cells[:cnt].cpu().numpy().view(np.uint64)
print(json.dumps({"n": n, "fam": fam, "keys": int(D.sk_off[-1]), "cells": int(cnt), "checksum": int(np.bitwise_xor.reduce(c * np.uint64(0x9E3779B97F4A7C15))), "touch": os.environ.get("SPSP_DEBUG_ACC_TOUCH", "default"),
                  "wall_ms": round(wall, 4), "pipeline_ms": round(t["compare_ms"] / 10, 4), "accumulate_ms": round(t["accumulate_ms"] / max(1, t["accumulate_launches"]), 4)}))
