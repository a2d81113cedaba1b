#!/usr/bin/env python3
"""Per-stage wall times of the sketching pipeline for one 5 Mbp genome (single thread, through the C-ABI)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

g = synth.random_genome(np.random.default_rng(1), 5_000_000)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "g.fa")
open(path, "wb").write(synth.to_fasta(g, "g", n_records=2))
ctx = sp.Context(0)
p = sp.make_params(31, 11, 1000)


def timed(f, reps=5):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r


ms, text = timed(lambda: sp.read_file(path)); print("read_file            %7.2f ms" % ms)
ms, (bases, off) = timed(lambda: sp.clean_fasta(text)); print("clean (host)         %7.2f ms" % ms)
ms, em = timed(lambda: ctx.scan(p, bases, off)); print("scan (H2D+GPU+D2H)   %7.2f ms  (%d super-k-mers)" % (ms, len(em)))
ms, (payload, st) = timed(lambda: sp.sketch_build(p, 1000.0, bases, off, em)); print("sketch_build (host)  %7.2f ms  (%d B)" % (ms, len(payload)))
ms, _ = timed(lambda: sp.write_gz(os.path.join(tmp, "o.gz"), payload, 9)); print("write_gz level 9    %7.2f ms" % ms)
ms, _ = timed(lambda: ctx.sketch_text(text, 31, 11, 1000.0)); print("sketch_text (GPU ingest, all-in-one) %7.2f ms" % ms)
ms, _ = timed(lambda: ctx.sketch_file(path, os.path.join(tmp, "o2.gz"))); print("sketch_file (whole)  %7.2f ms" % ms)
