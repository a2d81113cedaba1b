# what a 20-step run (the driver's) pays over a 200-step one, by factor: device keys, slots, batches
for cfg in "1 4 3" "0 4 3" "0 2 3" "0 2 1" "0 4 1"; do set -- $cfg
for st in 20 200; do
BENCH_DEVICE_KEYS=$1 BENCH_SLOTS=$2 BENCH_BATCHES=$3 timeout -k 10 120 python bench.py --experiment --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/m.json 2> gpurun_out/m.err && python -c "
import json
d=json.loads(open('gpurun_out/m.json').read().strip().splitlines()[-1]); print('device_keys $1 slots $2 batches $3 steps $st: %.4f ms/step dense %.4f' % (d['ms_per_step'], d['stage_ms']['dense_kernel']))"
done; done
