# analysis: the dense pass beside n empty launches per step on the small CUs (no comparison, no key extraction)
for n in 0 4 8 16 32; do
BENCH_DEBUG_NOISE_KERNELS=$n BENCH_DEVICE_KEYS=0 BENCH_DEBUG_SKIP_COMPARE=1 timeout -k 10 120 python bench.py --experiment --gpus 1 --steps 200 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/nz.json 2> gpurun_out/nz.err && python -c "
import json
d=json.loads(open('gpurun_out/nz.json').read().strip().splitlines()[-1]); print('noise kernels $n per step: step %.4f dense %.4f host queueing %.3f' % (d['ms_per_step'], d['stage_ms']['dense_kernel'], d['host_ms_per_step'].get('queueing', 0)))"
done
