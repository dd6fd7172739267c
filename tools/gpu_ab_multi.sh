# sweep an environment variable over values on one box: bash tools/gpu_ab_multi.sh VAR v1 v2 v3 ...
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$VAR=$v', round(d['value']), round(d['ms_per_step']*1e3,1))"
  done
done
