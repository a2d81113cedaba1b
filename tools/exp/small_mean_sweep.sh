# experiment: records per part of the small-problem comparison (k_parts_group_small: cap 2048, product mean 1450 = 315 parts for bench.py's batch)
for mean in 1450 1600 1750 1900; do for rep in 1 2; do
SPSP_DEBUG_SMALL_MEAN=$mean timeout -k 10 120 python bench.py --gpus 1 --steps 200 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/sm.json 2> gpurun_out/sm.err && python -c "
import json
d=json.loads(open('gpurun_out/sm.json').read().strip().splitlines()[-1]); print('small mean $mean: closed %.4f open %.4f dense %.4f nonzero %s' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel'], d['inter_nonzero']))"
done; done
