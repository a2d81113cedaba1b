#!/usr/bin/env python3
"""sender side of the key-partitioned split alone: spsp_partition_keys_device of one rank's block at BASELINE configs[3], G = 8
(1 250 sketches, ~6 x 10^6 keys -> 8 slots), host wall clock per call.  usage: tools/exp/part_bench.py [reps=50]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import supersampler_amd as sp
from supersampler_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
D = synth.direct_family_sketches(1250, fam_size=20, seed=4, device=dev, skm_range=(120, 360))
ctx = sp.Context(0)
G, per = 8, 1250
keys = int(D.sk_off[-1])
cap = int(keys / G * 1.25) + 4096
slot = sp.slot_bytes(per, cap, 31)
send = torch.zeros(G * slot, dtype=torch.uint8, device=dev)
call = lambda: ctx.partition_keys_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, per, G, cap, send.data_ptr())
for _ in range(5):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    call()
torch.cuda.synchronize()
print("partition of %d keys into %d slots: %.4f ms per call" % (keys, G, (time.perf_counter() - t0) * 1e3 / reps))
