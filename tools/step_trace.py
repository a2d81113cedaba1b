#!/usr/bin/env python3
"""Host-side timeline of the single-GPU pipelined step of bench.py (two contexts / two streams).
usage (GPU box): python tools/step_trace.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    K, M, S = 31, 11, 1000.0
    n_gen, glen = 100, int(os.environ.get("GLEN", "5000000"))
    wait = os.environ.get("WAIT_DENSE", "1") == "1"
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
    torch.cuda.synchronize()
    names = ["scan_begin", "wait_dense", "compare_begin", "scan_end", "compare_end"]
    acc = np.zeros(len(names))
    n_steps = 50
    for it in range(n_steps + 5):
        ts = [time.perf_counter()]
        ctx.scan_device_begin(p, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), n_gen)
        ts.append(time.perf_counter())
        if wait:
            ctx_cmp.wait_dense(ctx)
        ts.append(time.perf_counter())
        ctx_cmp.compare_device_begin(K, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n_gen, 0, 1, d_inter.data_ptr())
        ts.append(time.perf_counter())
        ctx.scan_device_end()
        ts.append(time.perf_counter())
        ctx_cmp.compare_end()
        ts.append(time.perf_counter())
        if it >= 5:
            acc += np.diff(ts)
    torch.cuda.synchronize()
    for n, a in zip(names, acc / n_steps * 1e6):
        print("%-24s %8.1f us" % (n, a))
    print("%-24s %8.1f us" % ("step", acc.sum() / n_steps * 1e6))


if __name__ == "__main__":
    main()
