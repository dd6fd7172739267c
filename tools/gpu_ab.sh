# A/B of an environment toggle on the bench (graph replay), alternating runs on the same box
for i in 1 2 3; do
  for v in 0 1; do
    env $1=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$1=$v', round(d['value']), round(d['ms_per_step']*1e3,1))"
  done
done
