timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_samplenet.py -q -m gpu -k "qrot or pcrnet or task or registration" 2>&1 | tail -5
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', round(d['value']), d['ms_per_step']); print(json.dumps(d['module_surface'])[:900])"
