#!/usr/bin/env python3
"""how open + read + close of 10 000 small files on tmpfs scales with threads (one directory / one directory per thread):
what bounds the first pass of spsp_compare_files's reader"""
import os, time, threading, shutil
root = "/dev/shm/open_scaling"
shutil.rmtree(root, ignore_errors=True)
n = 10000
for layout in ("one_dir", "dir_per_16"):
    os.makedirs(root, exist_ok=True)
    paths = []
    for i in range(n):
        d = root if layout == "one_dir" else os.path.join(root, "d%02d" % (i % 16))
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, "f%05d" % i)
        with open(p, "wb") as f: f.write(b"x" * 3200)
        paths.append(p)
    for T in (1, 2, 4, 8, 16):
        def work(t):
            for i in range(t, n, T) if layout == "one_dir" else [j for j in range(n) if j % 16 % T == t]:
                fd = os.open(paths[i], os.O_RDONLY); os.read(fd, 65536); os.close(fd)
        best = 1e9
        for _ in range(3):
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            best = min(best, time.perf_counter() - t0)
        print("%s threads %2d: %.2f ms for %d files" % (layout, T, best * 1e3, n))
    shutil.rmtree(root, ignore_errors=True)
