#!/bin/bash
# Where a sub_sampler process spends its wall time on N files (GPU box): HIP start-up floor (tools/exp/exp_hipstart),
# the pipeline's own report (SPSP_DEBUG_PIPE_TIMES) and the wall clock, three runs each.
# usage: bash tools/cli_startup.sh [n_files=100] [threads=16]
set -e
n=${1:-100}; t=${2:-16}
mkdir -p gpurun_out
d=$(mktemp -d /dev/shm/spsp_cli_XXXX)
trap 'rm -rf "$d"' EXIT
python - "$n" "$d" <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
from supersampler_amd import synth
n, d = int(sys.argv[1]), sys.argv[2]
gs = synth.family_genomes(2, n, 5_000_000, 10, [0.001, 0.01])
with open(os.path.join(d, "fof.txt"), "w") as fof:
    for i, g in enumerate(gs):
        p = os.path.join(d, "g%03d.fa" % i)
        open(p, "wb").write(synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3))
        fof.write(p + "\n")
P
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/exp/exp_hipstart.hip -o gpurun_out/hipstart      # (/dev/shm is mounted noexec)
for r in 1 2 3; do gpurun_out/hipstart; echo; done
cd "$d"
for r in 1 2 3; do
  s=$(date +%s%N)
  SPSP_DEBUG_PIPE_TIMES=1 "$OLDPWD/bin/sub_sampler" -f fof.txt -k 31 -m 11 -s 1000 -t "$t" -p "o${r}_" > run$r.out 2> run$r.err
  e=$(date +%s%N)
  echo "run $r: wall $(( (e - s) / 1000000 )) ms; $(grep -h 'spsp pipeline' run$r.err)"
done
