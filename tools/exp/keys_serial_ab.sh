# experiment: SPSP_KEYS_SERIAL=1 against the product's key extraction: parity tests, the tandem-repeat debug case, the step
mkdir -p gpurun_out
SPSP_KEYS_SERIAL=1 timeout -k 10 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "keys or smoke" > gpurun_out/ks_tests.log 2>&1; tail -2 gpurun_out/ks_tests.log
for i in 1 2 3; do SPSP_KEYS_SERIAL=1 python tools/exp/dbg_keys.py 21 11 3.0 2 2>&1 | grep "genome [34]" | awk '{print $6}' | tr "\n" " "; done; echo
for r in 1 2 3; do for ser in 0 1; do
SPSP_KEYS_SERIAL=$ser timeout -k 10 120 python bench.py --gpus 1 --steps 200 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ks.json 2> gpurun_out/ks.err && python -c "
import json
d=json.loads(open('gpurun_out/ks.json').read().strip().splitlines()[-1]); print('serial $ser: closed %.4f open %.4f dense %.4f parity %s' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel'], d['device_keys'].get('equal_to_keys_parsed_from_sketch_payloads')))"
done; done
