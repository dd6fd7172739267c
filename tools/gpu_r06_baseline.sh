# Round 6 baseline at the round's first sources: the GPU suite, the default bench line (all legs), nothing profiled.
# Run as: gpurun --timeout 1200 -- 'bash tools/gpu_r06_baseline.sh'
R=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
T="timeout -s KILL"
S0=$(date +%s)
$T 600 python -m pytest tests -m gpu -x -q --durations=10 > $OUT/pytest_gpu_baseline.log 2>&1
tail -15 $OUT/pytest_gpu_baseline.log
echo "tests done at $(( $(date +%s) - S0 )) s"
$T 300 python bench.py --steps 300 --warmup 30 > $OUT/bench_baseline.json 2> $OUT/bench_baseline.err
tail -3 $OUT/bench_baseline.err
python - <<PY
import json
d = json.loads(open('$OUT/bench_baseline.json').read().strip().splitlines()[-1])
print('headline', round(d['value']), round(d['ms_per_step'] * 1e3, 1), 'us')
for k in ('config1_classification', 'config3_sampler', 'config3_emd', 'config5_progressive', 'module_surface', 'batch_sweep', 'cpu_baseline'):
    print(k, json.dumps(d.get(k))[:1500])
PY
echo "bench done at $(( $(date +%s) - S0 )) s"
