#!/usr/bin/env python3
"""One LARGE FASTA file (a whole eukaryote genome: few records of hundreds of Mbp) through the file path
(spsp_sketch_files: pinned slab -> device ingest -> scan -> sketch payload -> gzip), checked against an independent
path over the same bases: spsp_scan_device + spsp_sketch_keys_device per record, keys compared with the decoded sketch.
usage: tools/exp/big_file.py [total_gbp=1.0] [records=4] [s=1000] [k=31] [m=11]"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp  # noqa: E402

total = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 1_000_000_000
n_rec = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
K = int(sys.argv[4]) if len(sys.argv) > 4 else 31
M = int(sys.argv[5]) if len(sys.argv) > 5 else 11
W = 60
rec_len = total // n_rec // W * W
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(11)
tmp = tempfile.mkdtemp(prefix="spsp_big_")
fa = os.path.join(tmp, "big.fa")
t0 = time.perf_counter()
recs = []
with open(fa, "wb") as f:
    for r in range(n_rec):
        b = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (rec_len,), generator=g, device=dev)]
        recs.append(b)
        lines = torch.cat([b.view(-1, W), torch.full((rec_len // W, 1), 10, dtype=torch.uint8, device=dev)], 1).cpu().numpy()
        f.write(b">chr%d\n" % r)
        f.write(lines.tobytes())
        del lines
print("wrote %s: %.2f GB in %.1f s" % (fa, os.path.getsize(fa) / 1e9, time.perf_counter() - t0), flush=True)
out = os.path.join(tmp, "big.sketch.gz")
t0 = time.perf_counter()
res, stages, _ = sp.sketch_files([fa], [out], k=K, m=M, s=s, threads=4)
wall = time.perf_counter() - t0
rc, stats, err = res[0]
assert rc == 0, err
print(json.dumps({"sketch_files_wall_s": round(wall, 3), "bases_per_s": round(rec_len * n_rec / wall), "stages": {k: round(v, 3) for k, v in stages.items() if v}, "stats": stats}), flush=True)
t0 = time.perf_counter()
res, stages, _ = sp.sketch_files([fa], [out], k=K, m=M, s=s, threads=4)
print("second call: %.3f s" % (time.perf_counter() - t0), flush=True)
# independent path: scan + keys per record on the device, one genome = all records
ctx = sp.Context(0)
p = sp.make_params(K, M, s)
bases = torch.cat(recs)
del recs
off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * rec_len
d_sk, n_sk = ctx.scan_device(p, bases.data_ptr(), bases.numel(), off.data_ptr(), n_rec)
d_mn, d_lo, d_hi, koff = ctx.sketch_keys_device(p, bases.data_ptr(), bases.numel(), off.data_ptr(), d_sk, n_sk, np.array([0, n_rec], dtype=np.uint32))
nk = int(koff[1])
import ctypes
mn = np.empty(nk, np.uint32); lo = np.empty(nk, np.uint64)
t_mn = torch.empty(nk, dtype=torch.int32, device=dev); t_lo = torch.empty(nk, dtype=torch.int64, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(ctypes.c_void_p(t_mn.data_ptr()), ctypes.c_void_p(d_mn), ctypes.c_size_t(nk * 4), 3)
hip.hipMemcpy(ctypes.c_void_p(t_lo.data_ptr()), ctypes.c_void_p(d_lo), ctypes.c_size_t(nk * 8), 3)
t_hi = torch.zeros(nk, dtype=torch.int64, device=dev)
if K > 32:
    hip.hipMemcpy(ctypes.c_void_p(t_hi.data_ptr()), ctypes.c_void_p(d_hi), ctypes.c_size_t(nk * 8), 3)
want = torch.stack([t_mn.to(torch.int64) & 0xffffffff, t_hi, t_lo], 1)
ctx2 = sp.Context(0)
payload = gzip.decompress(open(out, "rb").read())
k2, m2, e_mn, e_lo, e_hi, eoff = ctx2.sketch_decode_device([payload])
ne = int(eoff[1])
g_mn = torch.empty(ne, dtype=torch.int32, device=dev); g_lo = torch.empty(ne, dtype=torch.int64, device=dev)
hip.hipMemcpy(ctypes.c_void_p(g_mn.data_ptr()), ctypes.c_void_p(e_mn), ctypes.c_size_t(ne * 4), 3)
hip.hipMemcpy(ctypes.c_void_p(g_lo.data_ptr()), ctypes.c_void_p(e_lo), ctypes.c_size_t(ne * 8), 3)
g_hi = torch.zeros(ne, dtype=torch.int64, device=dev)
if K > 32:
    hip.hipMemcpy(ctypes.c_void_p(g_hi.data_ptr()), ctypes.c_void_p(e_hi), ctypes.c_size_t(ne * 8), 3)
got = torch.stack([g_mn.to(torch.int64) & 0xffffffff, g_hi, g_lo], 1)
same = ne == nk and bool((got == want).all())
print(json.dumps({"superkmers": int(n_sk), "keys_device_path": nk, "keys_in_sketch_file": ne, "identical": same, "sketch_bytes": os.path.getsize(out)}), flush=True)
os.remove(fa); os.remove(out); os.rmdir(tmp)
sys.exit(0 if same else 1)
