#!/bin/bash
# open + read + close (+ fstat, + inflate) of 10 000 small gzip files on tmpfs from T threads, raw system calls: how the first pass of
# spsp_compare_files's reader scales on this host.  usage: bash tools/exp/open_scaling.sh
set -e
R=$(pwd)
python3 - <<'P'
import os, gzip, numpy as np
os.makedirs("/dev/shm/rdtest", exist_ok=True)
rng = np.random.default_rng(1)
for i in range(10000):
    open("/dev/shm/rdtest/%05d.gz" % i, "wb").write(gzip.compress(b"51 11\n" + rng.integers(0, 256, 4000, dtype=np.uint8).tobytes(), 6))
P
g++ -O2 -o /tmp/open_scaling $R/tools/exp/open_scaling.cpp -lz -lpthread
for mode in 0 1 2; do for t in 1 2 4 8 16 32; do /tmp/open_scaling $t $mode | tail -1; done; done
rm -rf /dev/shm/rdtest
