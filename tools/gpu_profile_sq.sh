# SQ-counter passes over the conv stack's forward / fused-backward GEMM kernels (matrix-pipe busy share, LDS bank conflicts, waits) and the
# kernel stats of the secondary legs, bounded (run as: gpurun --timeout 240 -- 'bash tools/gpu_profile_sq.sh r05').  Counter passes carry
# --kernel-trace only (no --stats, no other trace domain).  tools/assemble_profile.sh reduces sq_fwd_* / sq_bwd_* to
# profiles/<round>/linear_fwd_sq_counters.json / conv_bwd_sq_counters.json.
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
T="timeout -s KILL"
S0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1))
  $T 40 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqf_$i -o p -- python $GRAFT_REPO_ROOT/tools/fwd_loop.py > /tmp/sqf_$i.log 2>&1
  cp /tmp/sqf_$i/p_counter_collection.csv $OUT/sq_fwd_$i.csv 2>/dev/null || tail -5 /tmp/sqf_$i.log
  $T 40 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqb_$i -o p -- python $GRAFT_REPO_ROOT/tools/bwd_loop.py > /tmp/sqb_$i.log 2>&1
  cp /tmp/sqb_$i/p_counter_collection.csv $OUT/sq_bwd_$i.csv 2>/dev/null || tail -5 /tmp/sqb_$i.log
done
echo "sq passes done at $(( $(date +%s) - S0 )) s"
$T 40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o v -- python $GRAFT_REPO_ROOT/tools/variant_loop.py config3_sampler 60 > /tmp/prof_c3.log 2>&1
cp /tmp/prof_c3/v_kernel_stats.csv $OUT/config3_sampler_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_c3.log
$T 40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eager -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-probes --no-graph > /tmp/prof_eager.log 2>&1
cp /tmp/prof_eager/bench_kernel_stats.csv $OUT/bench_eager_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_eager.log
$T 40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_task -o t -- python $GRAFT_REPO_ROOT/tools/task_loop.py > /tmp/prof_task.log 2>&1
cp /tmp/prof_task/t_kernel_stats.csv $OUT/task_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_task.log
echo "all done at $(( $(date +%s) - S0 )) s"
ls $OUT
