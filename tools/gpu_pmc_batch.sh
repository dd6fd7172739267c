# HBM counters of the whole step at a large batch (VERDICT r4 missing #4): separate FETCH_SIZE / WRITE_SIZE passes over the eager
# launches of bench.py --batch B.  Run as: gpurun --timeout 900 -- 'bash tools/gpu_pmc_batch.sh r05 2048 512'
R=${1:-r05}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  mkdir -p /tmp/pmcb_$b
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_${b}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 6 --warmup 3 --no-probes --no-graph --min-time 0 > /tmp/pmcb_${b}_$c.log 2>&1 || tail -5 /tmp/pmcb_${b}_$c.log
    cp /tmp/pmcb_${b}_$c/p_counter_collection.csv /tmp/pmcb_$b/pmc_$c.csv
  done
  python $GRAFT_REPO_ROOT/tools/summarize_pmc.py /tmp/pmcb_$b $OUT/pmc_summary_b$b.json
done
