#!/usr/bin/env python3
"""spsp_compare_files over N sketch files on tmpfs: GPU bulk decode (default) against the host decoder (SPSP_HOST_DECODE=1,
set before the library is first used), with the library's stage times.
usage (GPU box): python tools/decode_bench.py [N=1000]"""
import gzip
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
k, m, s = 31, 11, 50.0
rng = np.random.default_rng(3)
tmp = tempfile.mkdtemp(prefix="spsp_dec_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    ctx = sp.Context(0)
    p = sp.make_params(k, m, s)
    fam = max(1, N // 20)
    paths = []
    for f in range(fam):
        anc = synth.random_genome(rng, int(rng.integers(100_000, 400_000)))
        for j in range((N + fam - 1) // fam):
            if len(paths) >= N:
                break
            b, o = synth.concat_records([synth.mutate(rng, anc, [0.001, 0.01, 0.05][j % 3])])
            pl, _ = sp.sketch_build(p, s, b, o, ctx.scan(p, b, o))
            path = os.path.join(tmp, "s%05d.gz" % len(paths))
            open(path, "wb").write(gzip.compress(pl, 1))
            paths.append(path)
    ctx.compare_files(paths[:4], os.path.join(tmp, "warm"))
    ctx.stage_times(reset=True)
    t0 = time.perf_counter()
    ctx.compare_files(paths, os.path.join(tmp, "res"))
    wall = time.perf_counter() - t0
    st = ctx.stage_times(reset=True)
    print("%s decode: %d sketches, compare_files %.3f s: load(read+gunzip%s) %.3f, decode+compare+D2H %.3f, csv %.3f, csv gzip %.3f"
          % ("host" if os.environ.get("SPSP_HOST_DECODE") else "GPU", N, wall, "+decode+sort" if os.environ.get("SPSP_HOST_DECODE") else "",
             st["load_s"], st["compare_s"], st["csv_s"], st["csv_gzip_s"]))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
