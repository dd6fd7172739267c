# Round-6 measurement bundle, hard-bounded (run as: gpurun --timeout 1500 -- 'bash tools/gpu_profile_r06.sh r06'): every step under
# `timeout -s KILL`, the artefacts the bench line's roofline / traffic / profile.stale fields are read from first.  Everything lands in
# gpurun_out/<round>/; tools/assemble_profile.sh copies what is to be judged into profiles/<round>/.
R=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
T="timeout -s KILL"
S0=$(date +%s)
python -c "import bench; print(bench.csrc_sha16())" > $OUT/PROFILE_SRC_SHA
cd /tmp && export TMPDIR=/tmp
B="--no-probes"
$T 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_graph -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B > /tmp/prof_graph.log 2>&1
cp /tmp/prof_graph/bench_kernel_stats.csv $OUT/bench_graph_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  $T 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $B --no-graph > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
done
$T 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eager -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B --no-graph > /tmp/prof_eager.log 2>&1
cp /tmp/prof_eager/bench_kernel_stats.csv $OUT/bench_eager_kernel_stats.csv
for b in 512 2048; do
  $T 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o bench -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 30 --warmup 10 --no-probes > /tmp/prof_b$b.log 2>&1
  cp /tmp/prof_b$b/bench_kernel_stats.csv $OUT/b${b}_kernel_stats.csv
done
echo "core passes done at $(( $(date +%s) - S0 )) s"
cd $GRAFT_REPO_ROOT
$T 200 bash tools/gpu_pmc_batch.sh $R 2048 512 > $OUT/pmc_batch.log 2>&1
cd /tmp
for form in emd_loss three_call; do
  $T 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_emd_$form -o emd -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/prof_emd.log 2>&1
  cp /tmp/prof_emd_$form/emd_kernel_stats.csv $OUT/emd_${form}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    $T 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmce_${form}_$c -o p -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/pmce.log 2>&1
    cp /tmp/pmce_${form}_$c/p_counter_collection.csv $OUT/emd_${form}_pmc_$c.csv
  done
  $T 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sqe_$form -o p -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/sqe.log 2>&1
  cp /tmp/sqe_$form/p_counter_collection.csv $OUT/emd_${form}_sq.csv 2>/dev/null
done
echo "emd passes done at $(( $(date +%s) - S0 )) s"
# kernel stats of the secondary legs (config3_sampler K = 16, config5_progressive, configs[1]'s classification sampler, the task term)
$T 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o v -- python $GRAFT_REPO_ROOT/tools/variant_loop.py config3_sampler 60 > /tmp/prof_c3.log 2>&1
cp /tmp/prof_c3/v_kernel_stats.csv $OUT/config3_sampler_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_c3.log
$T 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o v -- python $GRAFT_REPO_ROOT/tools/variant_loop.py config5_progressive 40 > /tmp/prof_c5.log 2>&1
cp /tmp/prof_c5/v_kernel_stats.csv $OUT/config5_progressive_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_c5.log
$T 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o v -- python $GRAFT_REPO_ROOT/tools/cls_loop.py 200 7 > /tmp/prof_c1.log 2>&1
cp /tmp/prof_c1/v_kernel_stats.csv $OUT/config1_classification_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_c1.log
$T 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_task -o t -- python $GRAFT_REPO_ROOT/tools/task_loop.py > /tmp/prof_task.log 2>&1
cp /tmp/prof_task/t_kernel_stats.csv $OUT/task_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_task.log
$T 90 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/graph_trace.py task 30 > /tmp/tr.log 2>&1
{ grep "task_only" /tmp/tr.log; python $GRAFT_REPO_ROOT/tools/graph_trace.py --analyze /tmp/tr/t_kernel_trace.csv; } > $OUT/task_graph_timeline.txt 2>&1
echo "leg stats done at $(( $(date +%s) - S0 )) s"
cd $GRAFT_REPO_ROOT
$T 60 python tools/summarize_emd.py $OUT $OUT 3 > /dev/null 2>&1
$T 100 python tools/emd_bench.py > $OUT/emd_bench.txt 2>&1
$T 100 python tools/pairscan_scaling.py > $OUT/pairscan_scaling.txt 2>&1
$T 100 python tools/batch_sweep.py 32 128 512 2048 > $OUT/batch_sweep.txt 2>/dev/null
$T 200 python tools/surface_bench.py > $OUT/surface_bench.json 2>/dev/null
$T 100 python tools/surface_profile.py 500 > $OUT/surface_profile.txt 2>&1
# the module surface under data parallelism: reducer attached, RCCL all-reduce (forced at world size 1) inside the backward graph
$T 200 python bench.py --gpus 1 --force-collective --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs 2> $OUT/bench_n1_rccl_surface.err | tail -1 > $OUT/bench_n1_rccl_surface.json
for mode in auto graph after; do
  $T 120 python bench.py --gpus 1 --force-collective --allreduce $mode --steps 1500 --warmup 100 --no-probes 2> $OUT/bench_n1_rccl_$mode.err | tail -1 > $OUT/bench_n1_rccl_$mode.json
done
$T 120 python bench.py --steps 1500 --warmup 100 --no-probes 2>/dev/null | tail -1 > $OUT/bench_n1_noprobes.json
{
  echo "# tools/cotenancy_stress.py on one MI355X: two processes at once (parent + child), every pass compared with the process's first pass bit for bit"
  echo "## product build (compiler-packed fp32 off; emd.o: hand-written packed instructions, destinations disjoint from their sources)"
  $T 120 python tools/cotenancy_stress.py fwd 6000 2>&1 | grep cotenancy_stress
  $T 150 python tools/cotenancy_stress.py step 60 2>&1 | grep cotenancy_stress
  $T 150 python tools/cotenancy_stress.py task 25 2>&1 | grep cotenancy_stress
  $T 120 python tools/cotenancy_stress.py emd 4000 2>&1 | grep cotenancy_stress
  $T 120 python tools/cotenancy_stress.py scan 2000 2>&1 | grep cotenancy_stress
} > $OUT/cotenancy_stress.txt 2>&1
echo "side legs done at $(( $(date +%s) - S0 )) s"
# the bench line last, against THIS run's kernel stats (assembled on the box first: roofline / profile.stale refer to the profile it is committed with)
bash tools/assemble_profile.sh $R > /dev/null 2>&1
$T 500 python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 300 $OUT/bench_n1.json; echo
echo "all done at $(( $(date +%s) - S0 )) s"
ls $OUT | wc -l
