#!/bin/bash
# A/B of the timed step: this tree's library against the one built from an older commit (tools/exp/ab/libspsp_<commit>.so),
# alternating, on one box.  usage: tools/exp/ab/step_ab.sh <old.so> [rounds=3]
# the older library (built here, travels with the snapshot; *.so is git-ignored):
#   mkdir /tmp/old && git archive <commit> supersampler_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/supersampler_amd/csrc \
#     && cp /tmp/old/supersampler_amd/libspsp.so tools/exp/ab/libspsp_<commit>.so
# (cells_ab.py / host_ab.py / tools/rank_share.py / tools/c4_compare.py take the same SPSP_LIB=<old.so>)
old=$1; n=${2:-3}
for r in $(seq 1 $n); do
  for lib in new $old; do
    if [ $lib = new ]; then unset SPSP_LIB; else export SPSP_LIB=$lib; fi
    python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], round(d['ms_per_step'],5), round(d['stage_ms']['dense_kernel'],5), round(d['roofline']['frac'],4))"
  done
done
