# The parts of tools/gpu_profile_final.sh that tools/gpu_profile_core.sh / gpu_profile_sq.sh leave out, bounded
# (run as: gpurun --timeout 200 -- 'bash tools/gpu_profile_rest.sh r05 [a|b]'):
#   a: HBM counters of the whole step at B = 2048 / 512, the EMD kernels' stats / counters, pair-scan scaling, EMD bench, batch sweep
#   b: the module-surface legs, the RCCL placements at world size 1 (the two-process co-tenancy runs stay with gpu_profile_final.sh)
R=${1:-r05}
PART=${2:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
T="timeout -s KILL"
S0=$(date +%s)
cd $GRAFT_REPO_ROOT
if [ "$PART" = a ]; then
  $T 100 bash tools/gpu_pmc_batch.sh $R 2048 512 > $OUT/pmc_batch.log 2>&1
  echo "pmc batch done at $(( $(date +%s) - S0 )) s"
  cd /tmp && export TMPDIR=/tmp
  for form in emd_loss three_call; do
    $T 40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_emd_$form -o emd -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/prof_emd.log 2>&1
    cp /tmp/prof_emd_$form/emd_kernel_stats.csv $OUT/emd_${form}_kernel_stats.csv
    for c in FETCH_SIZE WRITE_SIZE; do
      $T 40 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmce_${form}_$c -o p -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/pmce.log 2>&1
      cp /tmp/pmce_${form}_$c/p_counter_collection.csv $OUT/emd_${form}_pmc_$c.csv
    done
    $T 40 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sqe_$form -o p -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/sqe.log 2>&1
    cp /tmp/sqe_$form/p_counter_collection.csv $OUT/emd_${form}_sq.csv 2>/dev/null
  done
  cd $GRAFT_REPO_ROOT
  $T 30 python tools/summarize_emd.py $OUT $OUT 3 > /dev/null 2>&1
  echo "emd passes done at $(( $(date +%s) - S0 )) s"
  $T 40 python tools/emd_bench.py > $OUT/emd_bench.txt 2>&1
  $T 40 python tools/pairscan_scaling.py > $OUT/pairscan_scaling.txt 2>&1
  $T 60 python tools/batch_sweep.py 32 128 512 2048 > $OUT/batch_sweep.txt 2>/dev/null
else
  $T 100 python tools/surface_bench.py > $OUT/surface_bench.json 2>/dev/null
  echo "surface bench done at $(( $(date +%s) - S0 )) s"
  $T 60 python bench.py --gpus 1 --force-collective --steps 300 --warmup 30 --no-cpu-baseline --no-extra-legs 2> $OUT/bench_n1_rccl_surface.err | tail -1 > $OUT/bench_n1_rccl_surface.json
  for mode in graph after; do
    $T 40 python bench.py --gpus 1 --force-collective --allreduce $mode --steps 1500 --warmup 100 --no-probes 2> $OUT/bench_n1_rccl_$mode.err | tail -1 > $OUT/bench_n1_rccl_$mode.json
  done
  echo "rccl legs done at $(( $(date +%s) - S0 )) s"
fi
echo "all done at $(( $(date +%s) - S0 )) s"
ls $OUT | wc -l
