cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mlp.py -x -q --timeout 900 -k "trainable_pcrnet or skinny" 2>&1 | tail -2
mkdir -p gpurun_out/r05; python bench.py --steps 300 --warmup 30 > gpurun_out/r05/bench_try.json 2> gpurun_out/r05/bench_try.err; tail -3 gpurun_out/r05/bench_try.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_try.json'))
print('value',round(d['value']),d['ms_per_step'])
for k in ('roofline','roofline_heaviest','roofline_step','roofline_geometry','b512_clouds_per_s','b2048_clouds_per_s','b2048_hbm_frac_algorithmic','config3_emd','module_surface','config3_sampler','config5_progressive'):
    print(k, json.dumps(d.get(k))[:700])
PY
