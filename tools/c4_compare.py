#!/usr/bin/env python3
"""BASELINE configs[3] on one GPU, alone (for rocprofv3 passes): N sketches synthesised directly
(synth.direct_family_sketches, the workload of bench.py's `compare_c4` leg), `reps` calls of spsp_compare_device.
usage: tools/c4_compare.py [N=10000] [reps=5]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
D = synth.direct_family_sketches(n, fam_size=20, seed=4, device=dev, skm_range=(120, 360))
d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
ctx = sp.Context(0)
call = lambda: ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, d_inter.data_ptr())  # noqa: E731
for _ in range(2):
    call()
ctx.timing_enable(True, sp.TIME_ALL)
ctx.timing_read()
for _ in range(reps):
    call()
t = ctx.timing_read()
ms = t["compare_ms"] / reps
print(json.dumps({"n": n, "keys": int(D.sk_off[-1]), "pipeline_ms": ms, "pairs_per_s": n * (n - 1) / 2 / (ms / 1e3),
                  "scatter_ms": t["scatter_ms"] / max(1, t["scatter_launches"]), "group_ms": t["group_ms"] / max(1, t["group_launches"]),
                  "accumulate_ms": t["accumulate_ms"] / max(1, t["accumulate_launches"]),
                  "nonzero_pairs": int(torch.count_nonzero(torch.triu(d_inter, 1)).item())}))
ctx.close()
