set -e
run() { name=$1; shift; env "$@" BENCH_FORCE_DIST=1 python bench.py --experiment --steps 400 --no-extras --no-cpu-baseline $EXTRA > gpurun_out/dv_$name.json 2> gpurun_out/dv_$name.err; python - <<P
import json
d=json.loads(open("gpurun_out/dv_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["ms_per_step"],4), d["host_ms_per_step"], d["config"]["exchange_check"][:20])
P
}
EXTRA=""
run shared2 BENCH_SLOTS=2
run cmp2 BENCH_SLOTS=2 BENCH_SMALL_STREAMS=cmp
run cmp3 BENCH_SLOTS=3 BENCH_SMALL_STREAMS=cmp
run shared3 BENCH_SLOTS=3
run cmp4 BENCH_SLOTS=4 BENCH_SMALL_STREAMS=cmp
EXTRA="--genomes 200 --length 2500000"
run g200_shared2 BENCH_SLOTS=2
run g200_cmp3 BENCH_SLOTS=3 BENCH_SMALL_STREAMS=cmp
run g200_cmp4 BENCH_SLOTS=4 BENCH_SMALL_STREAMS=cmp
run g200_cmp3_s96 BENCH_SLOTS=3 BENCH_SMALL_STREAMS=cmp BENCH_SMALL_CUS=96
