timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', round(d['value']), d['ms_per_step']); print(json.dumps(d['module_surface'])[:900]); print(json.dumps(d['config5_progressive'])[-260:])"
