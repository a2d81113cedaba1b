# experiment: the key extraction's stream on all 64 small CUs (product) or on 32 of them
for cus in "0,64" "32,32" "0,32"; do for st in 200; do
BENCH_KEYS_CUS=$cus timeout -k 10 120 python bench.py --experiment --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/kc.json 2> gpurun_out/kc.err && python -c "
import json
d=json.loads(open('gpurun_out/kc.json').read().strip().splitlines()[-1]); print('keys CUs $cus steps $st: closed %.4f open %.4f dense %.4f' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel']))"
done; done
