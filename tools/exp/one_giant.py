#!/usr/bin/env python3
"""configs[3]'s 10 000 sketches plus ONE giant sketch (a eukaryote among bacteria: G keys, a tenth of them taken from the
other sketches): the comparison with a row a thousand times longer than the others.  usage: tools/exp/one_giant.py [G=3000000]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
n0 = 10000
dev = torch.device("cuda", 0)
D = synth.direct_family_sketches(n0, fam_size=20, seed=4, device=dev, skm_range=(120, 360))
S0 = int(D.sk_off[-1])
g = torch.Generator(device=dev)
g.manual_seed(9)
take = torch.randint(0, S0, (G // 10,), generator=g, device=dev)
mn = torch.cat([D.minimizer[take], torch.randint(0, 4 ** 11, (G - G // 10,), generator=g, device=dev, dtype=torch.int32)])
lo = torch.cat([D.kmer_lo[take], torch.randint(0, 2 ** 62, (G - G // 10,), generator=g, device=dev, dtype=torch.int64)])
key = torch.unique(torch.stack([mn.to(torch.int64), lo], 1), dim=0)          # sorted by (minimizer, k-mer), distinct
for where in ("last", "first"):
    if where == "last":
        all_mn = torch.cat([D.minimizer, key[:, 0].to(torch.int32)]).contiguous()
        all_lo = torch.cat([D.kmer_lo, key[:, 1]]).contiguous()
        off = np.concatenate([D.sk_off, [D.sk_off[-1] + key.shape[0]]]).astype(np.uint64)
        giant = n0
    else:
        all_mn = torch.cat([key[:, 0].to(torch.int32), D.minimizer]).contiguous()
        all_lo = torch.cat([key[:, 1], D.kmer_lo]).contiguous()
        off = np.concatenate([[0], D.sk_off + key.shape[0]]).astype(np.uint64)
        giant = 0
    n = n0 + 1
    d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx = sp.Context(0)
    ts = []
    for r in range(4):
        t0 = time.perf_counter()
        ctx.compare_device(31, all_mn.data_ptr(), all_lo.data_ptr(), None, off, n, 0, 1, d_inter.data_ptr())
        ts.append((time.perf_counter() - t0) * 1e3)
    # the giant's row / column against set algebra for 30 sketches
    rng = np.random.default_rng(1)
    wrong = 0
    K = torch.stack([all_mn.to(torch.int64), all_lo], 1)
    o = off.astype(np.int64)
    gk = K[o[giant]:o[giant + 1]]
    for j in rng.integers(0, n, 30):
        j = int(j)
        if j == giant:
            continue
        other = K[o[j]:o[j + 1]]
        want = gk.shape[0] + other.shape[0] - torch.cat([gk, other]).unique(dim=0).shape[0]
        got = int(d_inter[min(j, giant), max(j, giant)].item())
        wrong += got != want
    print(json.dumps({"giant_keys": int(key.shape[0]), "giant_is": where, "compare_ms": [round(t, 2) for t in ts], "wrong_of_30": wrong}), flush=True)
    ctx.close()
