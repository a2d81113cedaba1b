for rep in 1 2; do for sl in 2 3 4 6; do
BENCH_SLOTS=$sl timeout -k 10 120 python bench.py --experiment --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/j_$sl.json 2> gpurun_out/j_$sl.err && python -c "
import json
d=json.loads(open('gpurun_out/j_$sl.json').read().strip().splitlines()[-1]); print('slots $sl: closed %.4f open %.4f dense %.4f' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel']))"
done; done
