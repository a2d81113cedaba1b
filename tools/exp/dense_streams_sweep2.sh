for ds in 2; do for sl in 6 8; do for st in 20 200; do
BENCH_SLOTS=$sl BENCH_DENSE_STREAMS=$ds timeout -k 10 120 python bench.py --experiment --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/k2_${sl}_$st.json 2> gpurun_out/k2_${sl}_$st.err && python -c "
import json
d=json.loads(open('gpurun_out/k2_${sl}_$st.json').read().strip().splitlines()[-1]); print('dense streams $ds slots $sl steps $st: closed %.4f open %.4f dense %.4f host %s' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel'], {k: round(v, 4) for k, v in d['host_ms_per_step'].items() if isinstance(v, float)}))"
done; done; done
