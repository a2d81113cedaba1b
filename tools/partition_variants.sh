#!/bin/bash
# A/B of the step's CU partition and small-stream layout on ONE box (bench.py, no extras).  usage: bash tools/partition_variants.sh
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --experiment --steps 600 --no-extras --no-cpu-baseline > gpurun_out/pv_$name.json 2> gpurun_out/pv_$name.err || { echo "$name failed"; tail -3 gpurun_out/pv_$name.err; return; }
python - <<P
import json
d=json.loads(open("gpurun_out/pv_$name.json").read().strip().splitlines()[-1]); print("%-22s %.4f ms/step  dense %.4f  host %s" % ("$name", d["ms_per_step"], d["stage_ms"]["dense_kernel"], {k: round(v, 3) for k, v in d["host_ms_per_step"].items()}))
P
}
run s64_shared_2 BENCH_SMALL_CUS=64
run s32_shared_2 BENCH_SMALL_CUS=32
run s32_cmp_2 BENCH_SMALL_CUS=32 BENCH_SMALL_STREAMS=cmp
run s32_cmp_3 BENCH_SMALL_CUS=32 BENCH_SMALL_STREAMS=cmp BENCH_SLOTS=3
run s32_own_3 BENCH_SMALL_CUS=32 BENCH_SMALL_STREAMS=own BENCH_SLOTS=3
run s32_own_4 BENCH_SMALL_CUS=32 BENCH_SMALL_STREAMS=own BENCH_SLOTS=4
run s64_shared_2b BENCH_SMALL_CUS=64
