# Round-end measurement bundle: bench JSON (with cpu_baseline), rocprofv3 kernel stats (graph replay and eager launches),
# PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE in their own runs).  Writes under gpurun_out/r01/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r01
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_graph -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /tmp/prof_graph.log 2>&1
cp /tmp/prof_graph/bench_kernel_stats.csv $OUT/bench_graph_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eager -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-graph > /tmp/prof_eager.log 2>&1
cp /tmp/prof_eager/bench_kernel_stats.csv $OUT/bench_eager_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
done
ls -la $OUT
