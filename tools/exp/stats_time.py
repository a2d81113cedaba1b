#!/usr/bin/env python3
"""spsp_count_superkmers_device on 6 x 5 Mbp random records (a batch of the file pipeline): wall per call; SPSP_DEBUG_STATS=chunks for the A/B"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
from supersampler_amd import synth
rng = np.random.default_rng(5)
recs = [synth.random_genome(rng, 5_000_000) for _ in range(6)]
bases, off = synth.concat_records(recs)
d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
d_o = torch.from_numpy(off.view(np.int64)).cuda()
ctx = sp.Context(0)
p = sp.make_params(31, 11, 1000)
got = [ctx.count_superkmers_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(recs)) for _ in range(3)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ctx.count_superkmers_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(recs))
dt = (time.perf_counter() - t0) / 10
print("mode %s: %d super-k-mers, %.3f ms per call (%.1f G positions/s)" % (os.environ.get("SPSP_DEBUG_STATS", "segments"), got[0], dt * 1e3, len(bases) / dt / 1e9))
