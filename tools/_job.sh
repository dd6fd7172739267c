cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -s KILL 400 python bench.py --steps 300 --warmup 30 > gpurun_out/r05/bench_n1_b.json 2> gpurun_out/r05/bench_n1_b.err; echo "bench rc=$?"; head -c 200 gpurun_out/r05/bench_n1_b.json
