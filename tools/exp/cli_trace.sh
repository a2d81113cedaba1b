#!/bin/bash
# one sub_sampler process on 100 x 5 Mbp files with the pipeline's task trace: where the wall clock goes
# usage: bash tools/exp/cli_trace.sh [threads=16] [slots=]
t=${1:-16}
SLOTS=$2
R=$(pwd)
d=$(mktemp -d /dev/shm/spsp_cli_XXXX)
trap 'rm -rf "$d"' EXIT
python - 100 "$d" <<'P'
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
cd "$d"
for r in 1 2; do
  s=$(date +%s%N)
  if [ -n "$SLOTS" ]; then export SPSP_DEBUG_PIPE_SLOTS=$SLOTS; fi
  SPSP_DEBUG_PIPE_TIMES=1 SPSP_DEBUG_PIPE_TRACE=1 "$R/bin/sub_sampler" -f fof.txt -k 31 -m 11 -s 1000 -t "$t" -p "o${r}_" > run$r.out 2> run$r.err
  e=$(date +%s%N)
  echo "run $r: wall $(( (e - s) / 1000000 )) ms; $(grep -h 'spsp pipeline' run$r.err); $(grep -h 'pipe\] call' run$r.err)"
  grep "pipe\] slot" run$r.err | awk '{ st=$4; if (!(st in first)) { first[st]=$5 } last[st]=$7; gsub(/[()]/, "", $9); n[st]++; sum[st]+=$9; if ($9 > mx[st]) mx[st]=$9 } END { for (s in first) printf "   stage %s: first begins %s ms, last ends %s ms; %d tasks, mean %.3f ms, longest %.3f ms\n", s, first[s], last[s], n[s], sum[s]/n[s], mx[s] }' | sort
done
