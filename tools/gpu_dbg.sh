timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 400 python bench.py --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/bench_dbg.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_dbg.json").read())
print(d['value'], d['ms_per_step'])
print({k:(round(v.get('ms_per_step'),4) if isinstance(v,dict) and 'ms_per_step' in v else None) for k,v in d['module_surface'].items()})
print(d['config5_progressive']['eager']['ms_per_step'], d['config5_progressive']['graph']['ms_per_step'])
PY
