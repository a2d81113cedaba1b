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
    # the same comparison as sparse cells (four column blocks per row at this width: the cells leave every block's row sums)
    cells = torch.zeros(1 << 25, dtype=torch.int64, device=dev)
    scratch = torch.zeros_like(d_inter)                          # (the API asks for n x n; at this size the cells leave the row sums directly and it stays zero)
    t0 = time.perf_counter()
    cnt = ctx.compare_cells_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())
    print("as cells: %d non-zero pairs in %.1f ms" % (cnt, (time.perf_counter() - t0) * 1e3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt2 = ctx.compare_cells_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())
    print("as cells, again: %d non-zero pairs in %.1f ms" % (cnt2, (time.perf_counter() - t0) * 1e3))
    nz = int(torch.count_nonzero(torch.triu(d_inter, 1)).item()) if n <= 20000 else None
    cw = cells[:cnt]
    ii, jj, cc = (cw >> 48) & 0xffff, (cw >> 32) & 0xffff, cw & 0xffffffff
    assert bool((d_inter[ii, jj].to(torch.int64) == cc).all()) and bool((ii < jj).all())
    sample_rows = rng.integers(0, n, size=20).tolist()
    for i in sample_rows:
        assert int((ii == i).sum().item()) == int(torch.count_nonzero(d_inter[i]).item())
    print("scratch matrix written: %s" % bool(scratch.any().item()))
    print("cells equal the dense matrix on every cell they name and on the non-zero counts of 20 sampled rows")
    try:
        ctx.compare_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, np.zeros(65538, np.uint64), 65536, 0, 1, d_inter.data_ptr())
        raise SystemExit("65536 sketches were accepted")
    except sp.SpspError as e:
        print("65536 sketches rejected:", str(e)[:90])


if __name__ == "__main__":
    main()
