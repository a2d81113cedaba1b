# closed step, 200 steps, three times (box-to-box and run-to-run spread)
for i in 1 2 3; do
  python bench.py --steps 200 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['stage_ms']['dense_kernel'], d['host_ms_per_step'])"
done
