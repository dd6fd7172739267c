# Round-end measurement bundle (run as: gpurun --timeout 2400 -- 'bash tools/gpu_profile.sh r05'): bench JSON (all legs), bench with the RCCL
# collective forced at world size 1 (inside the graph / after it), rocprofv3 kernel stats (graph replay and eager launches), PMC
# passes for HBM traffic (FETCH_SIZE / WRITE_SIZE in their own runs), SQ counters of the forward GEMMs, of the fused conv
# backward and of the EMD forms, the pair-scan batch sweep, the EMD timings, the FC-chain phase timeline, the task network's
# kernel stats.  Everything lands in gpurun_out/<round>/ -- copy what is to be judged into profiles/<round>/.
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import bench; print(bench.csrc_sha16())" > $OUT/PROFILE_SRC_SHA
for mode in graph after; do
  timeout 300 python bench.py --gpus 1 --force-collective --allreduce $mode --steps 1500 --warmup 100 --no-probes 2> $OUT/bench_n1_rccl_$mode.err | tail -1 > $OUT/bench_n1_rccl_$mode.json
  cat $OUT/bench_n1_rccl_$mode.json
done
timeout 300 python bench.py --steps 1500 --warmup 100 --no-probes 2>/dev/null | tail -1 > $OUT/bench_n1_noprobes.json; cat $OUT/bench_n1_noprobes.json
timeout 300 python tools/pairscan_scaling.py > $OUT/pairscan_scaling.txt 2>&1; cat $OUT/pairscan_scaling.txt
timeout 300 python tools/batch_sweep.py 32 128 512 2048 > $OUT/batch_sweep.txt 2>/dev/null; cat $OUT/batch_sweep.txt
timeout 300 python tools/surface_bench.py > $OUT/surface_bench.json 2>/dev/null; head -c 600 $OUT/surface_bench.json
timeout 200 python tools/surface_profile.py 500 2>/dev/null | grep -v "^ \|^$" > $OUT/surface_profile.txt; head -4 $OUT/surface_profile.txt
timeout 300 python tools/emd_bench.py > $OUT/emd_bench.txt 2>&1; tail -3 $OUT/emd_bench.txt
if [ -f tools/_ab/libsamplenet_hip_tl.so ]; then
  SAMPLENET_AMD_LIB=$PWD/tools/_ab/libsamplenet_hip_tl.so timeout 200 python tools/fc_chain_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/fc_chain_timeline.txt
  (python tools/timeline.py stack 32; python tools/timeline.py stack 512) 2>&1 | grep -A8 "5-layer" > $OUT/fwd_persist_timeline.txt
fi
cd /tmp && export TMPDIR=/tmp
B="--no-probes"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_graph -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B > /tmp/prof_graph.log 2>&1
cp /tmp/prof_graph/bench_kernel_stats.csv $OUT/bench_graph_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eager -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $B --no-graph > /tmp/prof_eager.log 2>&1
cp /tmp/prof_eager/bench_kernel_stats.csv $OUT/bench_eager_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_task -o task -- python $GRAFT_REPO_ROOT/tools/task_loop.py 100 > /tmp/prof_task.log 2>&1
cp /tmp/prof_task/task_kernel_stats.csv $OUT/task_kernel_stats.csv
# the secondary legs' captured steps and the large-batch steps (persistent forward GEMMs)
for leg in config3_sampler config5_progressive; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$leg -o leg -- python $GRAFT_REPO_ROOT/tools/variant_loop.py $leg 100 > /tmp/prof_$leg.log 2>&1
  cp /tmp/prof_$leg/leg_kernel_stats.csv $OUT/${leg}_kernel_stats.csv
done
for b in 512 2048; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o bench -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 30 --warmup 10 --no-probes > /tmp/prof_b$b.log 2>&1
  cp /tmp/prof_b$b/bench_kernel_stats.csv $OUT/b${b}_kernel_stats.csv
done
for form in emd_loss three_call; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_emd_$form -o emd -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/prof_emd.log 2>&1
  cp /tmp/prof_emd_$form/emd_kernel_stats.csv $OUT/emd_${form}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $B --no-graph > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
  for form in emd_loss three_call; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmce_${form}_$c -o p -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/pmce.log 2>&1
    cp /tmp/pmce_${form}_$c/p_counter_collection.csv $OUT/emd_${form}_pmc_$c.csv
  done
done
for form in emd_loss three_call; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sqe_$form -o p -- python $GRAFT_REPO_ROOT/tools/emd_loop.py $form 3 > /tmp/sqe.log 2>&1
  cp /tmp/sqe_$form/p_counter_collection.csv $OUT/emd_${form}_sq.csv 2>/dev/null || tail -5 /tmp/sqe.log
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqf_$i -o p -- python $GRAFT_REPO_ROOT/tools/fwd_loop.py > /tmp/sqf_$i.log 2>&1
  cp /tmp/sqf_$i/p_counter_collection.csv $OUT/sq_fwd_$i.csv 2>/dev/null || tail -5 /tmp/sqf_$i.log
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sqb_$i -o p -- python $GRAFT_REPO_ROOT/tools/bwd_loop.py > /tmp/sqb_$i.log 2>&1
  cp /tmp/sqb_$i/p_counter_collection.csv $OUT/sq_bwd_$i.csv 2>/dev/null || tail -5 /tmp/sqb_$i.log
done
cd $GRAFT_REPO_ROOT
# HBM counters of the whole step at the saturating batches (VERDICT r4 missing #4)
bash tools/gpu_pmc_batch.sh $R 2048 512 > $OUT/pmc_batch.log 2>&1
# results must not depend on what else runs on the GPU: two processes at once, and the stand-alone packed-fma probe
{
  echo "# tools/cotenancy_stress.py on one MI355X: two processes at once (parent + child), every pass compared with the process's first pass bit for bit"
  echo "## product build (compiler-packed fp32 off; emd.o: hand-written packed instructions, destinations disjoint from their sources)"
  timeout 300 python tools/cotenancy_stress.py fwd 8000 2>&1 | grep cotenancy_stress
  timeout 300 python tools/cotenancy_stress.py step 200 2>&1 | grep cotenancy_stress
  timeout 300 python tools/cotenancy_stress.py emd 6000 2>&1 | grep cotenancy_stress
  if [ -x tools/micro/pk_fma_cotenancy ]; then
    echo "## tools/micro/pk_fma_cotenancy (stand-alone: v_pk_fma_f32 with destination = source pair, exact-integer recurrence, every lane checked)"
    (cd tools/micro; ./pk_fma_cotenancy alias 3000 & ./pk_fma_cotenancy alias 3000; wait)
    (cd tools/micro; ./pk_fma_cotenancy plain 3000 & ./pk_fma_cotenancy plain 3000; wait)
  fi
} > $OUT/cotenancy_stress.txt 2>&1
python tools/summarize_emd.py $OUT $OUT 3
# the bench line last, against THIS run's kernel stats: assemble profiles/<round>/ on the box first so that the line's
# roofline_longest / profile.stale fields refer to the profile it is committed with
bash tools/assemble_profile.sh $R > /dev/null 2>&1
timeout 600 python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 300 $OUT/bench_n1.json; echo
ls -la $OUT
