#!/usr/bin/env python3
"""The comparator's upper limit: 65 535 sketches (the reference's uint32 pair key, Comparator.h:26) in one
spsp_compare_device call, device-resident 17 GB pair matrix, sampled rows checked against Python set algebra;
65 536 sketches must be rejected.  usage (GPU box): python tools/max_sketches.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n = 65535
    rng = np.random.default_rng(7)
    universe = rng.integers(1, 2**62, size=400_000, dtype=np.int64)
    fam = [np.sort(rng.choice(universe, size=40, replace=False)) for _ in range(3000)]
    sets = []
    for i in range(n):
        base = fam[i % 3000]
        keep = base[rng.random(40) < 0.85]
        sets.append(np.unique(keep).astype(np.uint64))
    counts = np.array([len(s) for s in sets])
    sk_off = np.zeros(n + 1, dtype=np.uint64)
    sk_off[1:] = np.cumsum(counts)
    lo = np.concatenate(sets)
    d_lo = torch.from_numpy(lo.view(np.int64)).to(dev)
    d_mn = torch.full((len(lo),), 3, dtype=torch.int32, device=dev)
    d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx = sp.Context(0)
    t0 = time.perf_counter()
    ctx.compare_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 0, 1, d_inter.data_ptr())
    torch.cuda.synchronize()
    print("%d sketches, %d keys: compare %.1f ms" % (n, len(lo), (time.perf_counter() - t0) * 1e3))
    bad = 0
    for i in rng.integers(0, n, size=40).tolist() + [0, n - 2]:
        row = d_inter[i].cpu().numpy()
        si = set(sets[i].tolist())
        for j in list(range(i + 1, n, 3000))[:25] + rng.integers(i + 1, n, size=25).tolist() if i + 1 < n else []:
            want = len(si & set(sets[j].tolist()))
            bad += int(row[j] != want)
        assert (row[: i + 1] == 0).all()
    print("sampled cells wrong: %d" % bad)
    assert bad == 0
    try:
        ctx.compare_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, np.zeros(65538, np.uint64), 65536, 0, 1, d_inter.data_ptr())
        raise SystemExit("65536 sketches were accepted")
    except sp.SpspError as e:
        print("65536 sketches rejected:", str(e)[:90])


if __name__ == "__main__":
    main()
