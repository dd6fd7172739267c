# SQ-counter passes over the conv stack's forward / fused-backward GEMM kernels at B = 32 AND at the saturating batch (B = 2048): matrix-pipe
# busy share, LDS bank conflicts, waits -> gpurun_out/<round>/sq_{fwd,bwd}_b<B>_<i>.csv; tools/assemble (below) reduces them to
# profiles/<round>/{conv_bwd,linear_fwd}_sq_counters[_b2048].json.  Run: gpurun --timeout 600 -- 'bash tools/gpu_sq_r06.sh r06'
R=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
T="timeout -s KILL"
cd /tmp && export TMPDIR=/tmp
for B in 32 2048; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY"; do
    i=$((i+1))
    SQ_BATCH=$B $T 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqf_${B}_$i -o p -- python $GRAFT_REPO_ROOT/tools/fwd_loop.py > /tmp/sqf.log 2>&1
    cp /tmp/sqf_${B}_$i/p_counter_collection.csv $OUT/sq_fwd_b${B}_$i.csv 2>/dev/null || tail -5 /tmp/sqf.log
    SQ_BATCH=$B $T 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqb_${B}_$i -o p -- python $GRAFT_REPO_ROOT/tools/bwd_loop.py > /tmp/sqb.log 2>&1
    cp /tmp/sqb_${B}_$i/p_counter_collection.csv $OUT/sq_bwd_b${B}_$i.csv 2>/dev/null || tail -5 /tmp/sqb.log
  done
done
ls $OUT | grep sq_
