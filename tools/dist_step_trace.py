#!/usr/bin/env python3
"""Host-side timeline of one pipelined multi-GPU step (world=1 rehearsal): how long each call keeps the host.
usage (GPU box): MASTER_ADDR=127.0.0.1 MASTER_PORT=29700 RANK=0 WORLD_SIZE=1 python tools/dist_step_trace.py"""
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
from supersampler_amd import synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    K, M, S = 31, 11, 1000.0
    n_gen, glen = 100, int(os.environ.get('GLEN', '5000000'))
    p = sp.make_params(K, M, S)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(sa)
    ctx, ctx_cmp = sp.Context(0, sa.cuda_stream), sp.Context(0, sb.cuda_stream)
    genomes = synth.family_genomes(2, n_gen, glen, 10, [0.001, 0.01])
    bases, rec_off = synth.concat_records(genomes)
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(rec_off.view(np.int64)).to(dev)
    rng = np.random.default_rng(1)
    counts = np.full(n_gen, 4500, dtype=np.int64)
    sk_off = np.zeros(n_gen + 1, dtype=np.uint64)
    sk_off[1:] = np.cumsum(counts)
    tot = int(sk_off[-1])
    lo = np.sort(rng.integers(0, 2**62, size=(n_gen, 4500), dtype=np.int64), axis=1).reshape(-1)
    d_lo = torch.from_numpy(lo).to(dev)
    d_min = torch.zeros(tot, dtype=torch.int32, device=dev)
    d_inter = torch.zeros((n_gen, n_gen), dtype=torch.int32, device=dev)
    ex = spd.SlotExchange(ctx_cmp, K, n_gen, tot, dev)
    torch.cuda.synchronize()
    names = ["scan_begin", "begin(partition+a2a)", "wait_dense", "end_queue", "scan_end", "compare_end", "all_reduce(launch)"]
    acc = np.zeros(len(names))
    n_steps = 50
    for it in range(n_steps + 5):
        ts = [time.perf_counter()]
        ctx.scan_device_begin(p, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), n_gen)
        ts.append(time.perf_counter())
        with torch.cuda.stream(sb):
            h = ex.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off)
        ts.append(time.perf_counter())
        ctx_cmp.wait_dense(ctx)
        ts.append(time.perf_counter())
        with torch.cuda.stream(sb):
            ex.end_queue(h, d_inter)
        ts.append(time.perf_counter())
        ctx.scan_device_end()
        ts.append(time.perf_counter())
        ctx_cmp.compare_end()
        ts.append(time.perf_counter())
        with torch.cuda.stream(sb):
            dist.all_reduce(d_inter)
        ts.append(time.perf_counter())
        if it >= 5:
            acc += np.diff(ts)
    torch.cuda.synchronize()
    for n, a in zip(names, acc / n_steps * 1e6):
        print("%-24s %8.1f us" % (n, a))
    print("%-24s %8.1f us" % ("step", acc.sum() / n_steps * 1e6))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
