cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1
grep -c "" $GRAFT_REPO_ROOT/gpurun_out/counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > /tmp/pmc_$name.log 2>&1
  ls /tmp/pmc_$name | head -3
  cp /tmp/pmc_$name/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.csv 2>/dev/null
done
python - <<'PY'
import csv, collections, glob, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+'pmc_*.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if not k.startswith(('void sn::','sn::','simp')): continue
    print(k)
    print('   ', {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
