# A/B on one box: bench.py's step with another build of the library (SPSP_LIB) against the in-tree one, alternating
# usage: bash tools/exp/ab_lib.sh <other libspsp.so> [steps=200] [reps=3]
other=$1; st=${2:-200}; reps=${3:-3}
for r in $(seq 1 $reps); do for which in tree other; do
  if [ $which = other ]; then export SPSP_LIB=$other; else unset SPSP_LIB; fi
  timeout -k 10 120 python bench.py --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err && python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$which steps $st: closed %.4f open %.4f dense %.4f' % (d['ms_per_step'], d['open_loop']['ms_per_step'], d['stage_ms']['dense_kernel']))"
done; done
