#!/usr/bin/env python3
"""BASELINE configs[4] (metagenome stress: 50 Gbp, k=63 m=15 s=100, records of 10^6 bp), streamed through HBM in
segments generated on the GPU (seeded; nothing is stored).  Every segment is one spsp_scan_device call; the
super-k-mer streams are checked for their invariants (ordered, disjoint per record, inside records) and the number
of selected k-mers against the expectation n/s.  Prints one JSON document.
C5_PACKED=1: the segments as 2-bit words (spsp_pack_bases_device; SPSP_SCAN_PACKED_INPUT), what the FASTA ingest hands the scan.
usage (GPU box): python tools/c5_scan.py [total_gbp=50] [segment_gbp=5]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402


def main():
    total_gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
    seg_gbp = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    dev = torch.device("cuda", 0)
    rec_len = 1_000_000
    seg_n = int(seg_gbp * 1e9) // rec_len * rec_len
    n_seg = max(1, int(round(total_gbp * 1e9 / seg_n)))
    n_rec = seg_n // rec_len
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    bases = torch.empty(seg_n + 64, dtype=torch.uint8, device=dev)
    off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * rec_len
    ctx = sp.Context(0)
    k, m, s = 63, 15, 100.0
    packed = os.environ.get("C5_PACKED") == "1"
    p = sp.make_params(k, m, s, flags=sp.SPSP_SCAN_PACKED_INPUT if packed else 0)
    ctx.timing_enable(True, sp.TIME_DENSE | sp.TIME_SCAN)
    segs = []
    tot_kmers = tot_sel = tot_sk = 0
    dense_ms = scan_ms = wall_s = 0.0
    for si in range(n_seg):
        step = 1 << 28
        for a in range(0, seg_n, step):
            b = min(seg_n, a + step)
            bases[a:b] = lut[torch.randint(0, 4, (b - a,), device=dev, generator=g, dtype=torch.int64)]
        torch.cuda.synchronize()
        src = ctx.pack_bases_device(bases.data_ptr(), seg_n) if packed else bases.data_ptr()
        torch.cuda.synchronize()
        if si == 0:
            ctx.scan_device(p, src, seg_n, off.data_ptr(), n_rec)      # warm-up: tables, buffers
            ctx.timing_read()
        t0 = time.perf_counter()
        d_out, n_out = ctx.scan_device(p, src, seg_n, off.data_ptr(), n_rec)
        wall = time.perf_counter() - t0
        t = ctx.timing_read()
        sk = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
        rec, start, ln = sk["rec"].astype(np.int64), sk["start"].astype(np.int64), sk["len"].astype(np.int64)
        assert (np.diff(rec) >= 0).all() and (ln >= k).all() and (start + ln <= rec_len).all() and (ln <= 2 * k - m).all()
        same = rec[1:] == rec[:-1]
        assert (start[1:][same] >= start[:-1][same] + ln[:-1][same] - k + 1).all()        # disjoint k-mer ranges, in order
        kmers = seg_n - n_rec * (k - 1)
        sel = int((ln - k + 1).sum())
        segs.append({"segment": si, "scan_ms": t["scan_ms"], "dense_ms": t["dense_ms"], "wall_ms": wall * 1e3,
                     "superkmers": int(n_out), "selected_kmers": sel})
        tot_kmers += kmers; tot_sel += sel; tot_sk += int(n_out)
        dense_ms += t["dense_ms"]; scan_ms += t["scan_ms"]; wall_s += wall
        print("segment %d/%d: %.1f Gbp scan %.2f ms (dense %.2f ms), %d super-k-mers, %d selected k-mers"
              % (si + 1, n_seg, seg_n / 1e9, t["scan_ms"], t["dense_ms"], n_out, sel), file=sys.stderr, flush=True)
    doc = {"input": "2-bit packed" if packed else "ASCII", "workload": "BASELINE configs[4]: %.0f Gbp as %d segments of %d records x 10^6 bp, k=63 m=15 s=100, generated on the GPU (seed 5)"
                       % (n_seg * seg_n / 1e9, n_seg, n_rec),
           "kmers": tot_kmers, "scan_pipeline_ms_total": scan_ms, "dense_kernel_ms_total": dense_ms, "host_wall_s_scan_calls": wall_s,
           "kmers_per_s_scan_pipeline": tot_kmers / (scan_ms / 1e3), "kmers_per_s_host_wall": tot_kmers / wall_s,
           "dense_kernel_GBps": n_seg * seg_n / 1e9 / (dense_ms / 1e3), "dense_kernel_frac_of_8TBps": n_seg * seg_n / 1e9 / (dense_ms / 1e3) / 8000.0,
           "superkmers": tot_sk, "selected_kmers": tot_sel, "expected_selected_kmers": tot_kmers / s,
           "selected_over_expected": tot_sel / (tot_kmers / s), "stream_invariants": "ordered, disjoint, inside records: OK on every segment",
           "segments": segs}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
