#!/usr/bin/env python3
"""N sketch files of ONE species through spsp_compare_files (cells path from 1024 files on) and spsp_compare_files_multi:
CSV bytes against the oracle's comparator + printers.  usage: tools/exp/species_files.py [N=1200] [genome=60000]"""
import gzip
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from supersampler_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
rng = np.random.default_rng(7)
anc = synth.random_genome(rng, L)
tmp = tempfile.mkdtemp(prefix="spsp_species_")
paths, payloads = [], []
t0 = time.perf_counter()
for i in range(n):
    g = synth.mutate(rng, anc, [0.0, 0.001, 0.003, 0.01][i % 4])
    pl = orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), 31, 11, 30.0)[0]
    pth = os.path.join(tmp, "sp_%04d.gz" % i)
    sp.write_gz(pth, pl, 1)
    paths.append(pth); payloads.append(pl)
print("%d sketches made in %.1f s" % (n, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
inter, card, _, _ = orc.compare(payloads)
print("oracle comparison %.1f s" % (time.perf_counter() - t0), flush=True)
want = {jac: orc.csv(jac, paths, inter, card, None, 6, 0.0) for jac in (True, False)}
ok = True
with sp.Context(0) as ctx:
    for rep in range(2):
        t0 = time.perf_counter()
        ctx.compare_files(paths, os.path.join(tmp, "one"))
        print("compare_files: %.3f s" % (time.perf_counter() - t0), {k: round(v, 4) for k, v in ctx.stage_times().items() if v}, flush=True)
for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
    ok &= gzip.open(os.path.join(tmp, "one") + suf, "rb").read() == want[jac]
# sortCSV over the dense Jaccard matrix: rows and columns into a shuffled file order, against the oracle's
jac_text = gzip.open(os.path.join(tmp, "one") + "_jaccard.csv.gz", "rb").read()
order = rng.permutation(n)
fof = ("\n".join(paths[i] for i in order) + "\n").encode()
t0 = time.perf_counter()
got_sorted = sp.sort_csv(jac_text, fof)
t1 = time.perf_counter()
want_sorted = orc.sort_csv(jac_text, fof)
print("sort_csv: %.3f s for %d MB (oracle %.1f s), equal: %s" % (t1 - t0, len(jac_text) >> 20, time.perf_counter() - t1, got_sorted == want_sorted), flush=True)
ok &= got_sorted == want_sorted
for rep in range(2):
    t0 = time.perf_counter()
    sp.compare_files_multi([0, 0, 0], paths, os.path.join(tmp, "multi"))
    print("compare_files_multi over 3 contexts: %.3f s" % (time.perf_counter() - t0), flush=True)
for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
    ok &= gzip.open(os.path.join(tmp, "multi") + suf, "rb").read() == want[jac]
print("CSV bytes equal the oracle's: %s" % ok)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
sys.exit(0 if ok else 1)
