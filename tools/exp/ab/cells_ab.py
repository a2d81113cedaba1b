#!/usr/bin/env python3
"""configs[3]'s comparison returned as cells, and configs[2]'s true-shape size synthesised directly (1 000 sketches): wall ms per call.
For A/B runs with SPSP_LIB=<older library>."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = sp.Context(0)
out = []
for n, rng in ((10000, (120, 360)), (1000, (100, 380))):
    D = synth.direct_family_sketches(n, fam_size=20, seed=4, device=dev, skm_range=rng)
    scratch = torch.zeros((n, n), dtype=torch.int32, device=dev)
    cells = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for what in ("cells", "dense"):
        if what == "cells":
            call = lambda: ctx.compare_cells_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())  # noqa: E731
        else:
            call = lambda: ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, scratch.data_ptr())  # noqa: E731
        for _ in range(3):
            call()
        t0 = time.perf_counter()
        for _ in range(32):
            call()
        out.append("%d %s %.3f" % (n, what, (time.perf_counter() - t0) * 1e3 / 32))
print(os.environ.get("SPSP_LIB", "new").split("/")[-1], " | ".join(out))
