#!/usr/bin/env python3
"""BASELINE configs[2] at its true shape on one GPU, alone (for rocprofv3 passes): N genomes of L ~ U[2, 8] Mbp in families of
20 generated on the device and sketched by the HIP path (bench.config3_true_shape: one scan + one key extraction per
batch of 100 genomes, SPSP_KEYS_UNORDERED), then `reps` calls of spsp_compare_device over the keys.
usage: tools/c3_compare.py [N=1000] [reps=10]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import supersampler_amd as sp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
ctx = sp.Context(0)
T = bench.config3_true_shape(ctx, dev, n, False, unordered=True)
ctx.compare_keys_unordered(True)
d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
call = lambda: ctx.compare_device(T["k"], T["d_min"].data_ptr(), T["d_lo"].data_ptr(), None, T["sk_off"], n, 0, 1, d_inter.data_ptr())  # noqa: E731
for _ in range(3):
    call()
ctx.timing_enable(True, sp.TIME_ALL)
ctx.timing_read()
for _ in range(reps):
    call()
t = ctx.timing_read()
ms = t["compare_ms"] / reps
print(json.dumps({"n": n, "keys": int(T["sk_off"][-1]), "bases": T["bases"], "genomes_through_the_table_in_hbm": T["big_genomes"],
                  "scan_ms_all_batches": T["scan_ms"], "sketch_keys_ms_all_batches": T["keys_ms"], "pipeline_ms": ms,
                  "pairs_per_s": n * (n - 1) / 2 / (ms / 1e3), "scatter_ms": t["scatter_ms"] / max(1, t["scatter_launches"]),
                  "group_ms": t["group_ms"] / max(1, t["group_launches"]), "accumulate_ms": t["accumulate_ms"] / max(1, t["accumulate_launches"]),
                  "nonzero_pairs": int(torch.count_nonzero(torch.triu(d_inter, 1)).item())}))
ctx.close()
