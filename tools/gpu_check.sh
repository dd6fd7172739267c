# pytest -m gpu + one graph-replay bench line.  Run as: gpurun -- 'bash tools/gpu_check.sh'
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 600 -s 2>&1 | grep -v "^\.*$" | tail -150 > gpurun_out/pytest_gpu.log
tail -120 gpurun_out/pytest_gpu.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_quick.json
python -c "import json; d=json.load(open('gpurun_out/bench_quick.json')); print('graph', d['value'], d['ms_per_step'])"
