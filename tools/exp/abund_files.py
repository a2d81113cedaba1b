#!/usr/bin/env python3
"""-a 2 over 100 FASTA files of 5 Mbp: the batched pipeline (one abundance pass per batch) against one GPU job per file
(SPSP_DEBUG_ABUND_PER_FILE=1); usage: tools/exp/abund_files.py [n_files=100] [threads=16]"""
import os, sys, time, tempfile, shutil, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import supersampler_amd as sp
from supersampler_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if os.environ.get("ABUND_CHILD"):
    ins = [l.strip() for l in open(os.environ["ABUND_CHILD"])]
    outs = [x + ".gz" for x in ins]
    sp.sketch_files(ins[:8], outs[:8], 31, 11, 1000.0, abundance=2, threads=T)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); res, st, _ = sp.sketch_files(ins, outs, 31, 11, 1000.0, abundance=2, threads=T); best = min(best, time.perf_counter() - t0)
        assert all(r[0] == 0 for r in res)
    print(json.dumps({"per_file": bool(os.environ.get("SPSP_DEBUG_ABUND_PER_FILE")), "files": len(ins), "threads": T, "wall_s": best, "files_per_s": len(ins) / best}))
    sys.exit(0)
tmp = tempfile.mkdtemp(prefix="spsp_ab_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    gs = synth.family_genomes(7, n, 5_000_000, 10, [0.001, 0.01])
    ins = []
    for i, g in enumerate(gs):
        pth = os.path.join(tmp, "g%03d.fa" % i)
        open(pth, "wb").write(synth.to_fasta(g, "g%d" % i))
        ins.append(pth)
    open(os.path.join(tmp, "fof"), "w").write("\n".join(ins) + "\n")
    for per_file in (True, False, True, False):
        env = dict(os.environ, ABUND_CHILD=os.path.join(tmp, "fof"))
        if per_file: env["SPSP_DEBUG_ABUND_PER_FILE"] = "1"
        print(subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(T)], env=env, capture_output=True, text=True).stdout.strip())
finally:
    shutil.rmtree(tmp, ignore_errors=True)
