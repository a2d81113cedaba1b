#!/usr/bin/env python3
"""All-vs-all comparison at BASELINE config-3 scale (N sketches of ~5000 k-mers, family structure):
GPU spsp_compare_device timing (+ optional oracle check).  Genomes are shortened and s lowered so that
each sketch still holds ~L/s ~ 5000 k-mers without generating 5 Gbp.

usage: tests/tools/compare_bench.py [N=1000] [check=1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
check = int(sys.argv[2]) if len(sys.argv) > 2 else 1
k, m, s = 31, 11, 50.0
rng = np.random.default_rng(3)
fam = max(1, N // 20)
t0 = time.time()
genomes = []
for f in range(fam):
    L = int(rng.integers(100_000, 400_000))          # "RefSeq-like spread" of sizes
    anc = synth.random_genome(rng, L)
    for j in range((N + fam - 1) // fam):
        if len(genomes) < N:
            genomes.append(synth.mutate(rng, anc, [0.001, 0.01, 0.05][j % 3]))
ctx = sp.Context(0, torch.cuda.current_stream().cuda_stream or None)
p = sp.make_params(k, m, s)
sketches, payloads = [], []
for g in genomes:
    b, o = synth.concat_records([g])
    em = ctx.scan(p, b, o)
    pl, _ = sp.sketch_build(p, s, b, o, em)
    payloads.append(pl)
    sketches.append(sp.sketch_parse(pl))
cnt = np.array([len(x) for x in sketches])
print("built %d sketches in %.1fs: keys/sketch mean %.0f (min %d max %d), total %d" % (N, time.time() - t0, cnt.mean(), cnt.min(), cnt.max(), cnt.sum()), flush=True)
dev = torch.device("cuda", 0)
d_min = torch.from_numpy(np.concatenate([x.minimizer for x in sketches]).view(np.int32)).to(dev)
d_lo = torch.from_numpy(np.concatenate([x.kmer_lo for x in sketches]).view(np.int64)).to(dev)
sk_off = np.zeros(N + 1, np.uint64); sk_off[1:] = np.cumsum(cnt)
d_inter = torch.zeros((N, N), dtype=torch.int32, device=dev)
torch.cuda.synchronize()   # torch fills on its own stream; the context has its own
for _ in range(2):
    ctx.compare_device(k, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, N, 0, 1, d_inter.data_ptr())
ctx.timing_enable(True); ctx.timing_read()
reps = 5
for _ in range(reps):
    ctx.compare_device(k, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, N, 0, 1, d_inter.data_ptr())
t = ctx.timing_read()
ms = t["compare_ms"] / reps
pairs = N * (N - 1) // 2
alg = 8.0 * (cnt.sum() * (N - 1)) + 4.0 * pairs      # sum over pairs of 8*(n_i+n_j)+4
print("compare pipeline %.3f ms (accumulate %.3f ms): %.3g pairs/s; no-reuse model %.1f GB -> %.0f GB/s (%.2fx of 8 TB/s); compulsory %.3f GB"
      % (ms, t["accumulate_ms"] / reps, pairs / ms * 1e3, alg / 1e9, alg / ms / 1e6, alg / ms / 1e6 / 8000, (8.0 * cnt.sum() + 4.0 * pairs) / 1e9), flush=True)
if check == 2:   # sampled pairs against numpy set intersections (the oracle's colour map is too heavy at N=10^4)
    got = d_inter.cpu().numpy().astype(np.uint32)
    keysets = None
    rs = np.random.default_rng(0)
    bad = 0
    comp = [sketches[i].minimizer.astype(np.uint64) << np.uint64(40) ^ sketches[i].kmer_lo * np.uint64(0x9E3779B97F4A7C15) for i in range(N)]
    for _ in range(400):
        i, j = sorted(int(x) for x in rs.integers(0, N, size=2))
        if i == j:
            continue
        a = np.stack([sketches[i].minimizer.astype(np.uint64), sketches[i].kmer_lo], 1)
        b = np.stack([sketches[j].minimizer.astype(np.uint64), sketches[j].kmer_lo], 1)
        sa = set(map(tuple, a.tolist())); sb = set(map(tuple, b.tolist()))
        if got[i, j] != len(sa & sb):
            bad += 1
    # plus every pair inside the first two families
    print("sampled parity: %d mismatches in 400 random pairs; nonzero pairs %d" % (bad, np.count_nonzero(np.triu(got, 1))), flush=True)
    sys.exit(0 if bad == 0 else 1)
if check:
    from oracle import oracle_py as orc
    t0 = time.time()
    want, card, sec = orc.compare(payloads, timed=True)
    got = d_inter.cpu().numpy().astype(np.uint32)
    ok = (np.triu(got, 1) == np.triu(want, 1)).all() and (card == cnt).all()
    print("oracle compare %.2fs (%.3g pairs/s single thread): parity %s, nonzero pairs %d" % (sec, pairs / sec, "OK" if ok else "MISMATCH", np.count_nonzero(want)), flush=True)
    sys.exit(0 if ok else 1)
