# Round-end measurement bundle (run as: gpurun -- 'bash tools/gpu_profile.sh r02'): bench JSON (with cpu_baseline and the
# module-surface leg), bench under torchrun with the RCCL collectives forced at world size 1, rocprofv3 kernel stats (graph
# replay and eager launches), PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE in their own runs), SQ counters of the
# forward GEMMs and of the fused conv backward, the pair-scan batch sweep, and the EMD kernels at BASELINE configs[3].
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 400 $OUT/bench_n1.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 \
    --steps 300 --warmup 30 --no-cpu-baseline --no-module-surface --force-collective > $OUT/bench_n1_rccl_forced.json 2> $OUT/bench_n1_rccl_forced.err
tail -c 300 $OUT/bench_n1_rccl_forced.json; echo
python tools/pairscan_scaling.py > $OUT/pairscan_scaling.txt 2>&1; cat $OUT/pairscan_scaling.txt
python tools/emd_bench.py > $OUT/emd_bench.txt 2>&1; cat $OUT/emd_bench.txt
cd /tmp && export TMPDIR=/tmp
B="--no-probes"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_graph -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B > /tmp/prof_graph.log 2>&1
cp /tmp/prof_graph/bench_kernel_stats.csv $OUT/bench_graph_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eager -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B --no-graph > /tmp/prof_eager.log 2>&1
cp /tmp/prof_eager/bench_kernel_stats.csv $OUT/bench_eager_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_emd -o emd -- python $GRAFT_REPO_ROOT/tools/emd_bench.py > /tmp/prof_emd.log 2>&1
cp /tmp/prof_emd/emd_kernel_stats.csv $OUT/emd_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $B --no-graph > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmce_$c -o p -- python $GRAFT_REPO_ROOT/tools/emd_bench.py > /tmp/pmce_$c.log 2>&1
  cp /tmp/pmce_$c/p_counter_collection.csv $OUT/emd_pmc_$c.csv
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqf_$i -o p -- python $GRAFT_REPO_ROOT/tools/fwd_loop.py > /tmp/sqf_$i.log 2>&1
  cp /tmp/sqf_$i/p_counter_collection.csv $OUT/sq_fwd_$i.csv 2>/dev/null || tail -5 /tmp/sqf_$i.log
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqb_$i -o p -- python $GRAFT_REPO_ROOT/tools/bwd_loop.py > /tmp/sqb_$i.log 2>&1
  cp /tmp/sqb_$i/p_counter_collection.csv $OUT/sq_bwd_$i.csv 2>/dev/null || tail -5 /tmp/sqb_$i.log
done
ls -la $OUT
