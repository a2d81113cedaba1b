#!/usr/bin/env python3
"""End-to-end sketching rate of spsp_sketch_files: N synthetic genomes as FASTA files on tmpfs -> sketch files, for a list
of worker counts; payloads of the first files checked against the oracle.  SPSP_FILES_PER_WORKER=1 in the environment
selects the one-GPU-job-per-file form (A/B).
usage: tools/e2e_files.py [n_files=100] [length=5000000] [threads=1,8,16] [reps=3]"""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
threads = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,8,16").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
K, M, S = 31, 11, 1000.0
tmp = tempfile.mkdtemp(prefix="spsp_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    gs = synth.family_genomes(2, n, length, 10, [0.001, 0.01])
    ins, texts = [], []
    for i, g in enumerate(gs):
        t = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
        pth = os.path.join(tmp, "g%03d.fa" % i)
        open(pth, "wb").write(t)
        ins.append(pth)
        texts.append(t)
    kmers = sum(len(g) - K + 1 for g in gs)
    outs = [os.path.join(tmp, "s%03d.gz" % i) for i in range(n)]
    sp.sketch_files(ins[:8], outs[:8], K, M, S, threads=8)      # HIP modules, page cache
    doc = {"files": n, "length": length, "kmers": kmers, "mode": "per-worker" if os.environ.get("SPSP_FILES_PER_WORKER") else "batched", "runs": {}}
    for T in threads:
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            res, st, _ = sp.sketch_files(ins, outs, K, M, S, threads=T)
            wall = time.perf_counter() - t0
            assert all(r[0] == 0 for r in res)
            if best is None or wall < best[0]:
                best = (wall, st)
        doc["runs"]["threads_%d" % T] = {"wall_s": best[0], "kmers_per_s": kmers / best[0],
                                         "stage_s": {k: best[1][k] for k in ("read_s", "ingest_s", "scan_s", "gather_s", "build_s", "gzip_s")}}
    from oracle import oracle_py as orc
    doc["parity_first_4"] = all(sp.read_file(outs[i]) == orc.sketch_fasta(texts[i], K, M, S)[0] for i in range(min(4, n)))
    print(json.dumps(doc))
finally:
    sp.sketch_files_release()
    shutil.rmtree(tmp, ignore_errors=True)
