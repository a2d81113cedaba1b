#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLIs on synthetic FASTA (host I/O + PCIe included).
usage: tools/cli_e2e.py [n_genomes=20] [len=5000000] [threads=8] [gzip_input=0]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from supersampler_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
gz = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tmp = tempfile.mkdtemp(prefix="spsp_e2e_")
gs = synth.family_genomes(2, n, L, max(1, n // 10), [0.001, 0.01])
names = []
for i, g in enumerate(gs):
    data = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
    p = os.path.join(tmp, "genome%03d.fa%s" % (i, ".gz" if gz else ""))
    open(p, "wb").write(gzip.compress(data, 1) if gz else data)
    names.append(p)
open(os.path.join(tmp, "fof.txt"), "w").write("\n".join(names) + "\n")
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", "fof.txt", "-t", str(thr), "-v", "0", "-p", "sk_"], cwd=tmp,
                   capture_output=True, text=True)
t1 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "sk_fof.txt", "-o", "res"], cwd=tmp, capture_output=True, text=True)
t2 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
kmers = sum(len(g) - 31 + 1 for g in gs)
print("sub_sampler: %d genomes x %d bp (%s input), %d threads: %.2f s -> %.3g k-mers/s end to end (incl. process start, HIP init, file I/O, gzip -9 output)"
      % (n, L, "gzip" if gz else "plain", thr, t1 - t0, kmers / (t1 - t0)))
print("comparator : %d sketches: %.2f s -> %.3g pairs/s end to end" % (n, t2 - t1, n * (n - 1) / 2 / (t2 - t1)))
