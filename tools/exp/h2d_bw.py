#!/usr/bin/env python3
"""H2D bandwidth from pinned memory: one stream, and two streams at once (is the 100-file pipeline's copy at the link's rate?)"""
import time, torch
dev = torch.device("cuda", 0)
for mb in (4, 16, 64, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(4, 1024 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    one = n * reps / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
    torch.cuda.synchronize()
    two = 2 * n * reps / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(reps): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    back = n * reps / (time.perf_counter() - t0) / 1e9
    print("%4d MiB: H2D one stream %.1f GB/s, two streams %.1f GB/s, D2H %.1f GB/s" % (mb, one, two, back))
