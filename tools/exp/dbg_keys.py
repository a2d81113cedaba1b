#!/usr/bin/env python3
"""debug: sorted vs unordered key extraction on the test genomes of test_sketch_keys_on_device_equal_the_file_path
usage: tools/exp/dbg_keys.py k m s ab"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import supersampler_amd as sp
from supersampler_amd import synth
k, m, s, ab = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1000 + k)
L = min(900_000, int((1500 if k > 32 else 3000) * s))
a = synth.random_genome(rng, L)
comp = {65: 84, 67: 71, 71: 67, 84: 65}
rc = np.array([comp[c] for c in a[::-1].tolist()], dtype=np.uint8)
unit = synth.random_genome(rng, k + 2)
genomes = [[a[: L // 2], a[L // 2:]], [synth.mutate(rng, a, 0.02)], [a[: L // 2], rc],
           [np.tile(unit, 257), synth.random_genome(rng, 500)], [np.tile(unit[::-1].copy(), 256)], [synth.random_genome(rng, k - 1)],
           [synth.random_genome(rng, L // 2)], [a[: L // 4]] * ab + [a[L // 4: L // 3]] * max(1, ab - 1)]
recs, first_rec = [], [0]
for g in genomes:
    recs += g
    first_rec.append(len(recs))
bases, off = synth.concat_records(recs)
d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
d_o = torch.from_numpy(off.view(np.int64)).cuda()
torch.cuda.synchronize()
ctx = sp.Context(0)
p = sp.make_params(k, m, s, abundance=ab)
d_sk, n_sk = ctx.scan_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(recs))
out = {}
for un in (False, True):
    d_mn, d_lo, d_hi, off2 = ctx.sketch_keys_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, first_rec, unordered=un)
    tot = int(off2[-1])
    mn, lo = ctx.to_host(d_mn, tot, np.uint32), ctx.to_host(d_lo, tot, np.uint64)
    out[un] = [sorted(zip(mn[int(off2[g]):int(off2[g + 1])].tolist(), lo[int(off2[g]):int(off2[g + 1])].tolist())) for g in range(len(genomes))]
for g in range(len(genomes)):
    A, B = set(out[False][g]), set(out[True][g])
    print("genome", g, "sorted", len(out[False][g]), "unordered", len(out[True][g]), "dups in unordered", len(out[True][g]) - len(B), "missing", len(A - B), "extra", len(B - A))
    for x in list(A - B)[:3]: print("   missing", x)
    for x in list(B - A)[:3]: print("   extra", x)
