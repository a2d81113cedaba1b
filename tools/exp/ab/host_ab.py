#!/usr/bin/env python3
"""host cost of a comparison call: 300 sketches of 60 keys (general partition form), 2000 calls, wall us per call.  For A/B with SPSP_LIB."""
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
for n in (300, 3000):
    D = synth.direct_family_sketches(n, fam_size=20, seed=4, device=dev, skm_range=(3, 4))
    d = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    call = lambda: ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, d.data_ptr())  # noqa: E731
    for _ in range(20):
        call()
    t0 = time.perf_counter()
    for _ in range(2000):
        call()
    print(os.environ.get("SPSP_LIB", "new").split("/")[-1], n, int(D.sk_off[-1]), "%.1f us per call" % ((time.perf_counter() - t0) * 1e6 / 2000))
