# pytest -m gpu + the default bench line + the all-reduce placement A/B at world size 1 (collective forced).
# Run as: gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 -x -s --durations=8 2>&1 | grep -v "^\.*$" | tail -150 > gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
python bench.py --steps 300 --warmup 30 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json
tail -5 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_default.json'))
print('headline', round(d['value']), round(d['ms_per_step'] * 1e3, 1), 'us')
for k in ('module_surface', 'config3_emd', 'config5_progressive', 'batch_sweep', 'roofline_longest'):
    print(k, json.dumps(d.get(k))[:900])
PY
for mode in after graph graph-fork after graph graph-fork; do
  python bench.py --gpus 1 --force-collective --allreduce $mode --steps 1000 --warmup 50 --no-probes 2>gpurun_out/ar_$mode.err | tail -1 > gpurun_out/ar_$mode.json
  echo "allreduce=$mode $(cat gpurun_out/ar_$mode.json | head -c 300)"; tail -2 gpurun_out/ar_$mode.err
done
python bench.py --steps 1000 --warmup 50 --no-probes | tail -1
