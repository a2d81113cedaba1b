#!/usr/bin/env python3
"""spsp_csv_cells_gz_host alone (host only): a 10 000 x 10 000 matrix of families of 20 as cells -> .csv.gz; seconds per matrix and a
check of the gunzipped bytes against the text form.  usage: tools/exp/csv_gz_time.py [n=10000]"""
import gzip, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(3)
names = ["/dev/shm/some/dir/s%05d.gz" % i for i in range(n)]
card = rng.integers(3000, 9000, n).astype(np.uint64)
cells = []
for f in range(n // 20):
    for a in range(20):
        for b in range(a + 1, 20):
            i, j = f * 20 + a, f * 20 + b
            cells.append((i << 48) | (j << 32) | int(rng.integers(1, 2500)))
cells = np.array(cells, dtype=np.uint64)
out = "/dev/shm/csv_gz_time.csv.gz"
for jac in (True, False):
    sp.csv_cells_gz(jac, names, cells, card, out)
    t0 = time.perf_counter()
    for _ in range(5): sp.csv_cells_gz(jac, names, cells, card, out)
    dt = (time.perf_counter() - t0) / 5
    ok = gzip.open(out, "rb").read() == sp.csv_cells(jac, names, cells, card) if n <= 4000 else None
    print("%s: %.2f ms per matrix, %d bytes, equal to the text form: %s" % ("jaccard" if jac else "containment", dt * 1e3, os.path.getsize(out), ok))
os.remove(out)
