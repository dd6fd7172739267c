# SQ counters of the pair scan at the saturating batch (B = 8192) and at B = 32: is the kernel VALU-issue-bound?
# run as: gpurun --timeout 90 -- 'bash tools/gpu_pairscan_sq.sh r05'  -> gpurun_out/<round>/pairscan_sq_*.csv
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -s KILL 40 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/psq_$i -o p -- python $GRAFT_REPO_ROOT/tools/pairscan_scaling.py 8192 32 > /tmp/psq_$i.log 2>&1
  cp /tmp/psq_$i/p_counter_collection.csv $OUT/pairscan_sq_$i.csv 2>/dev/null || tail -5 /tmp/psq_$i.log
done
ls $OUT | grep pairscan_sq
