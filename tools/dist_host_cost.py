#!/usr/bin/env python3
"""Host-side cost of the pieces of the multi-GPU step at one rank (RCCL with world 1): where do the ~0.13 ms of
queueing per step go?  usage (GPU box): python tools/dist_host_cost.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import dist as spd  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
h = sp.stream_create_cus(0, 0, 64)
st = torch.cuda.ExternalStream(h, device=dev)
ctx = sp.Context(0, st.cuda_stream)
n, per = 100, 4560
counts = np.full(n, per, dtype=np.int64)
rng = np.random.default_rng(1)
lo = np.sort(rng.integers(0, 2**62, size=(n, per), dtype=np.int64), axis=1).reshape(-1)
d_lo = torch.from_numpy(lo).to(dev)
d_mn = torch.zeros(n * per, dtype=torch.int32, device=dev)
d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
ex = spd.KeyExchange(counts, dev, stream=st)
torch.cuda.synchronize()
acc = {}


def timed(name, f):
    t0 = time.perf_counter()
    r = f()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


steps = 300
for it in range(steps + 20):
    if it == 20:
        acc.clear()
    g = timed("exchange (2 copies, 1 all_gather, 1 gather kernel)", lambda: ex.exchange(d_mn, d_lo))
    timed("compare_device_begin", lambda: ctx.compare_device_begin(31, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, ex.sk_off, n, 0, 1, d_inter.data_ptr()))
    timed("compare_end", lambda: ctx.compare_end())
    timed("collect_rows (world 1: nothing to move)", lambda: ex.collect_rows(d_inter))
    timed("  collect_rows' mask multiply alone", lambda: d_inter.mul_(ex._own_mask(d_inter)))
    timed("  one reduce alone", lambda: dist.reduce(d_inter, dst=0))
    with torch.cuda.stream(st):
        timed("  one all_gather_into_tensor alone", lambda: dist.all_gather_into_tensor(ex._g_buf, ex._pad_buf))
        timed("  the compaction gather alone", lambda: torch.index_select(ex._g_as_min, 0, ex._idx_all, out=ex._all_buf))
    def enter_leave():
        with torch.cuda.stream(st):
            pass
    timed("  entering + leaving torch.cuda.stream() alone", enter_leave)
torch.cuda.synchronize()
for k, v in acc.items():
    print("%-50s %7.1f us" % (k, v / steps * 1e6))
ctx.close()
dist.destroy_process_group()
