#!/usr/bin/env python3
"""End-to-end comparator CLI at config-3 scale: N sketches of ~5000 k-mers (shortened genomes, s=50)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from supersampler_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
tmp = tempfile.mkdtemp(prefix="spsp_e2e_")
rng = np.random.default_rng(3)
names = []
fam = max(1, N // 20)
i = 0
for f in range(fam):
    anc = synth.random_genome(rng, int(rng.integers(100_000, 400_000)))
    for j in range((N + fam - 1) // fam):
        if i < N:
            p = os.path.join(tmp, "g%04d.fa" % i)
            open(p, "wb").write(synth.to_fasta(synth.mutate(rng, anc, [0.001, 0.01, 0.05][j % 3]), "g%d" % i))
            names.append(p); i += 1
open(os.path.join(tmp, "fof.txt"), "w").write("\n".join(names) + "\n")
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", "fof.txt", "-t", "4", "-s", "50", "-v", "0", "-p", "sk_"], cwd=tmp, capture_output=True, text=True)
t1 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "sk_fof.txt", "-o", "res"], cwd=tmp, capture_output=True, text=True)
t2 = time.time()
assert r.returncode == 0, r.stdout + r.stderr
print("sub_sampler %d genomes: %.2f s; comparator: %.2f s -> %.3g pairs/s end to end (load+decode %d sketches, GPU compare, two %dx%d CSVs gzipped)"
      % (N, t1 - t0, t2 - t1, N * (N - 1) / 2 / (t2 - t1), N, N, N))
print(r.stdout.strip().split("\n")[-1])
