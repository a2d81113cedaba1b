#!/bin/bash
# What the comparison's kernels cost the dense pass they run beside (analysis flags; lines are not measurements):
# the step without its comparison, and with single stages of it skipped (SPSP_DEBUG_SKIP_STAGES: 1 scatter, 2 group),
# for 64 and 32 small CUs, on one box.
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --experiment --steps 600 --no-extras --no-cpu-baseline > gpurun_out/pb_$name.json 2> gpurun_out/pb_$name.err || { echo "$name failed"; return; }
python - <<P
import json
d=json.loads(open("gpurun_out/pb_$name.json").read().strip().splitlines()[-1]); print("%-22s %.4f ms/step  dense %.4f" % ("$name", d["ms_per_step"], d["stage_ms"]["dense_kernel"]))
P
}
for s in 64 32; do
run s${s} BENCH_SMALL_CUS=$s
run s${s}_nocompare BENCH_SMALL_CUS=$s BENCH_DEBUG_SKIP_COMPARE=1
run s${s}_prepare_only BENCH_SMALL_CUS=$s SPSP_DEBUG_SKIP_STAGES=3
run s${s}_scatter_only BENCH_SMALL_CUS=$s SPSP_DEBUG_SKIP_STAGES=2
run s${s}_group_only BENCH_SMALL_CUS=$s SPSP_DEBUG_SKIP_STAGES=1
done
