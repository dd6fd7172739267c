for i in 1 2; do
for v in 1 0; do
SAMPLENET_AMD_WGRAD_SIDE_STREAM=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('side=$v', round(d['value']), d['ms_per_step'])"
done; done
