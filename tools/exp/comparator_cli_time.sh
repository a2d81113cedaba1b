#!/bin/bash
# where a bin/comparator process on 100 sketch files spends its wall time: strace -c style is not here, so: the process's wall, the library's
# stage timers (SPSP_DEBUG_DECODE_TIMES), and the HIP start-up floor (tools/exp/exp_hipstart)
R=$(pwd)
d=$(mktemp -d /dev/shm/spsp_cmp_XXXX)
trap 'rm -rf "$d"' EXIT
python3 - "$d" <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
import supersampler_amd as sp
from supersampler_amd import synth
d = sys.argv[1]
D = synth.direct_family_sketches(100, fam_size=20, seed=4)
with open(os.path.join(d, "fof.txt"), "w") as f:
    for i in range(100):
        p = os.path.join(d, "s%03d.gz" % i); sp.write_gz(p, D.payload(i), 9); f.write(p + "\n")
P
cd "$d"
for r in 1 2 3; do
  s=$(date +%s%N)
  SPSP_DEBUG_DECODE_TIMES=1 "$R/bin/comparator" -f fof.txt -o res$r > out$r.txt 2> err$r.txt
  e=$(date +%s%N)
  echo "run $r: wall $(( (e - s) / 1000000 )) ms"; grep -h "spsp\|load\]" err$r.txt | head -5; tail -2 out$r.txt
done
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 $R/tools/exp/exp_hipstart.hip -o /tmp/hipstart 2>/dev/null && /tmp/hipstart
