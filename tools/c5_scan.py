#!/usr/bin/env python3
"""C5-shaped scan (BASELINE configs: k=63 m=15 s=100, records of 10^6 bp) at a size given in Gbp, generated on the GPU.
Prints the scan rate and checks the super-k-mer stream's invariants (ordered, disjoint per record, inside records).
usage (GPU box): python tools/c5_scan.py [gbp=8]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    dev = torch.device("cuda", 0)
    n = int(gbp * 1e9) // 16 * 16
    rec_len = 1_000_000
    n_rec = (n + rec_len - 1) // rec_len
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    bases = torch.empty(n, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        bases[a:b] = lut[torch.randint(0, 4, (b - a,), device=dev, generator=g, dtype=torch.int64)]
    off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * rec_len
    off[-1] = n
    torch.cuda.synchronize()
    ctx = sp.Context(0)
    k, m, s = 63, 15, 100.0
    p = sp.make_params(k, m, s)
    ctx.scan_device(p, bases.data_ptr(), n, off.data_ptr(), n_rec)      # warm-up: tables, buffers
    ctx.timing_enable(True, sp.TIME_DENSE | sp.TIME_SCAN)
    ctx.timing_read()
    t0 = time.perf_counter()
    d_out, n_out = ctx.scan_device(p, bases.data_ptr(), n, off.data_ptr(), n_rec)
    wall = time.perf_counter() - t0
    t = ctx.timing_read()
    kmers = n - n_rec * (k - 1)
    print("%.1f Gbp, %d records, k=%d m=%d s=%g: scan %.2f ms wall (pipeline %.2f ms, dense %.2f ms) -> %.3g k-mers/s; "
          "%d super-k-mers" % (n / 1e9, n_rec, k, m, s, wall * 1e3, t["scan_ms"], t["dense_ms"], kmers / wall, n_out))
    sk = ctx.to_host(d_out, min(n_out, 2_000_000), sp.SUPERKMER_DTYPE)
    rec, start, ln = sk["rec"].astype(np.int64), sk["start"].astype(np.int64), sk["len"].astype(np.int64)
    assert (np.diff(rec) >= 0).all() and (ln >= k).all() and (start + ln <= rec_len).all()
    same = rec[1:] == rec[:-1]
    assert (start[1:][same] > start[:-1][same]).all()
    print("stream invariants OK on the first %d super-k-mers; expected ~%.3g selected k-mers, got %.3g"
          % (len(sk), kmers / s, float((ln - k + 1).sum()) * n_out / max(1, len(sk))))


if __name__ == "__main__":
    main()
