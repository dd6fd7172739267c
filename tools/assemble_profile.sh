# gpurun_out/<round>/ (written by tools/gpu_profile.sh on the GPU box) -> profiles/<round>/: the small files as they are, the
# counter CSVs reduced to per-kernel summaries, and the kernel table.  Run from the repo root: bash tools/assemble_profile.sh r03
R=${1:-r06}
S=gpurun_out/$R
D=profiles/$R
mkdir -p $D
for f in PROFILE_SRC_SHA bench_n1.json bench_n1_noprobes.json bench_n1_rccl_graph.json bench_n1_rccl_after.json bench_graph_kernel_stats.csv \
         bench_eager_kernel_stats.csv task_kernel_stats.csv emd_emd_loss_kernel_stats.csv emd_three_call_kernel_stats.csv emd_bench.txt \
         emd_pmc_summary.json emd_sq_counters.json pairscan_scaling.txt fc_chain_timeline.txt pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv \
         config3_sampler_kernel_stats.csv config5_progressive_kernel_stats.csv b512_kernel_stats.csv b2048_kernel_stats.csv batch_sweep.txt \
         conv_bwd_sq_counters_b2048.json linear_fwd_sq_counters_b2048.json surface_bench.json surface_profile.txt fwd_persist_timeline.txt cotenancy_stress.txt pmc_summary_b2048.json pmc_summary_b512.json bench_n1_rccl_surface.json \
         bench_n1_rccl_auto.json config1_classification_kernel_stats.csv task_graph_timeline.txt; do
  [ -f $S/$f ] && cp $S/$f $D/$f
done
python tools/summarize_pmc.py $S $D/pmc_summary.json > /dev/null
# (the SQ-counter passes of the GEMM kernels are optional: tools/gpu_profile.sh runs them, the bounded tools/gpu_profile_final.sh does not)
ls $S/sq_bwd_*.csv > /dev/null 2>&1 && python tools/summarize_sq.py "$S/sq_bwd_*.csv" conv_bwd $D/conv_bwd_sq_counters.json > /dev/null
ls $S/sq_fwd_*.csv > /dev/null 2>&1 && python tools/summarize_sq.py "$S/sq_fwd_*.csv" linear_fwd $D/linear_fwd_sq_counters.json > /dev/null
python tools/kernel_table.py $R > /dev/null
python - <<PY
import bench
p = bench.profile_provenance()
print("profile provenance:", p)
PY
