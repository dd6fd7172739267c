# The core of the round-end bundle for a short GPU budget (run as: gpurun --timeout 420 -- 'bash tools/gpu_profile_core.sh r05'):
# what the bench line's `roofline` / `traffic` / `profile.stale` fields are read from, most important first, every step under
# `timeout -s KILL`; then the GPU test suite.  tools/gpu_profile_final.sh is the full bundle (~15 GPU-minutes).
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
T="timeout -s KILL"
S0=$(date +%s)
python -c "import bench; print(bench.csrc_sha16())" > $OUT/PROFILE_SRC_SHA
cd /tmp && export TMPDIR=/tmp
B="--no-probes"
$T 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_graph -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B > /tmp/prof_graph.log 2>&1
cp /tmp/prof_graph/bench_kernel_stats.csv $OUT/bench_graph_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  $T 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $B --no-graph > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
done
echo "profile passes done at $(( $(date +%s) - S0 )) s"
cd $GRAFT_REPO_ROOT
# the bench line against THIS run's kernel stats (assembled on the box first: roofline / profile.stale refer to the profile it is committed with)
bash tools/assemble_profile.sh $R > /dev/null 2>&1
$T 150 python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 200 $OUT/bench_n1.json; echo
echo "bench done at $(( $(date +%s) - S0 )) s"
$T 60 python bench.py --steps 1500 --warmup 100 --no-probes 2>/dev/null | tail -1 > $OUT/bench_n1_noprobes.json
cat $OUT/bench_n1_noprobes.json
# the GPU suite at these sources (the driver runs it again at round end)
$T 200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
echo "tests done at $(( $(date +%s) - S0 )) s"
cd /tmp
for b in 512 2048; do
  $T 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o bench -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 30 --warmup 10 --no-probes > /tmp/prof_b$b.log 2>&1
  cp /tmp/prof_b$b/bench_kernel_stats.csv $OUT/b${b}_kernel_stats.csv
done
echo "all done at $(( $(date +%s) - S0 )) s"
ls $OUT | wc -l
