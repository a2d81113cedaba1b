for rep in 1 2; do for dep in 4 3 2; do for st in 20 200; do
BENCH_COLLECT_DEPTH=$dep timeout -k 10 120 python bench.py --experiment --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/cd.json 2> gpurun_out/cd.err && python -c "
import json
d=json.loads(open('gpurun_out/cd.json').read().strip().splitlines()[-1]); print('collect depth $dep steps $st: closed %.4f open %.4f dense %.4f parity %s' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel'], d['device_keys'].get('equal_to_keys_parsed_from_sketch_payloads')))"
done; done; done
