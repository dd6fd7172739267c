cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -s KILL 400 python -X faulthandler bench.py --steps 300 --warmup 30 > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
echo "bench rc=$?"; tail -c 300 gpurun_out/r05/bench_n1.json; echo; grep -m3 "what()\|Fatal\|Error" gpurun_out/r05/bench_n1.err | cut -c1-200
timeout -s KILL 400 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -6
