#!/usr/bin/env python3
"""Throughput of the GPU ingest (spsp_fasta_clean_device) on a synthetic multi-record FASTA resident in HBM."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

n_g = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(1)
text = b"".join(synth.to_fasta(synth.random_genome(rng, 5_000_000), "g%d" % i, n_records=1 + i % 3) for i in range(n_g))
d = torch.from_numpy(np.frombuffer(text + b"\0" * 16, dtype=np.uint8).copy()).cuda()
torch.cuda.synchronize()
ctx = sp.Context(0)
ctx.clean_fasta_device(d.data_ptr(), len(text))
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    db, nb, do, nr = ctx.clean_fasta_device(d.data_ptr(), len(text))
dt = (time.perf_counter() - t0) / reps
print("GPU ingest: %d bytes of FASTA -> %d bases, %d records: %.3f ms per call (wall, 2 syncs) = %.0f GB/s of text"
      % (len(text), nb, nr, dt * 1e3, len(text) / dt / 1e9))
t0 = time.perf_counter()
b, o = sp.clean_fasta(text)
dt = time.perf_counter() - t0
print("host ingest (spsp_fasta_clean_host, 1 thread): %.1f ms = %.2f GB/s" % (dt * 1e3, len(text) / dt / 1e9))
