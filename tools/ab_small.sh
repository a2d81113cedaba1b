#!/bin/bash
# A/B of two builds of libspsp.so on ONE box: bench step at 64 and 32 small CUs.  usage: bash tools/ab_small.sh <old.so>
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --experiment --steps 600 --no-extras --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name failed"; tail -3 gpurun_out/ab_$name.err; return; }
python - <<P
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1]); print("%-14s %.4f ms/step  dense %.4f" % ("$name", d["ms_per_step"], d["stage_ms"]["dense_kernel"]))
P
}
OLD=$PWD/$1
run new_s64 BENCH_SMALL_CUS=64
run old_s64 BENCH_SMALL_CUS=64 SPSP_LIB=$OLD
run new_s32 BENCH_SMALL_CUS=32
run old_s32 BENCH_SMALL_CUS=32 SPSP_LIB=$OLD
run new_s64b BENCH_SMALL_CUS=64
run old_s64b BENCH_SMALL_CUS=64 SPSP_LIB=$OLD
