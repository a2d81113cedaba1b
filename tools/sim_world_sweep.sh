#!/bin/bash
# One rank's share of the comparison at world size W beside the usual scan, for several CU partitions (bench.py --experiment
# BENCH_SIM_WORLD; analysis only).  usage (GPU box): bash tools/sim_world_sweep.sh "2 4 8" "64 96 128 160 192"
set -e
mkdir -p gpurun_out
for w in ${1:-2 4 8}; do
  for s in ${2:-64 96 128 160 192}; do
    BENCH_SIM_WORLD=$w BENCH_SMALL_CUS=$s python bench.py --experiment --steps ${STEPS:-150} --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/sw_${w}_${s}.json 2> gpurun_out/sw_${w}_${s}.err
    python - <<P
import json
d=json.loads(open("gpurun_out/sw_${w}_${s}.json").read().strip().splitlines()[-1])
print("world $w small_cus $s: %.4f ms/step  dense %.4f  compare pipeline %s" % (d["ms_per_step"], d["stage_ms"]["dense_kernel"], d["stage_ms"]["compare_pipeline"]))
P
  done
done
