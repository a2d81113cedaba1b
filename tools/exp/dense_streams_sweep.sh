# experiment: consecutive dense passes on one stream (product) vs two unordered streams over the same CUs
for rep in 1 2; do for ds in 1 2; do for st in 20 200; do
BENCH_DENSE_STREAMS=$ds timeout -k 10 120 python bench.py --experiment --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/k_${ds}_$st.json 2> gpurun_out/k_${ds}_$st.err && python -c "
import json
d=json.loads(open('gpurun_out/k_${ds}_$st.json').read().strip().splitlines()[-1]); print('dense streams $ds steps $st: closed %.4f open %.4f dense %.4f parity %s' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel'], d['device_keys'].get('equal_to_keys_parsed_from_sketch_payloads')))"
done; done; done
