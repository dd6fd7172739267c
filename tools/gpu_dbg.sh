mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 -s 2>&1 | grep -v "^\.*$" | tail -150 > gpurun_out/pytest_gpu.log
grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3
grep -n "^FAILED\|^ERROR" gpurun_out/pytest_gpu.log | head -20
python -X faulthandler bench.py --steps 300 --warmup 30 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
tail -25 gpurun_out/bench_default.err | cut -c1-400
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), round(d['ms_per_step'] * 1e3, 1), 'us', d['timed_steps'], d['timed_region_s'])
for k in ('module_surface', 'config3_emd', 'config5_progressive', 'batch_sweep', 'roofline_longest', 'cpu_baseline'):
    print(k, json.dumps(d.get(k))[:1500])
PY
