cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcb_$i -o p -- python $GRAFT_REPO_ROOT/tools/bwd_loop.py > /tmp/pmcb_$i.log 2>&1
  cp /tmp/pmcb_$i/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmcb_$i.csv 2>/dev/null || tail -5 /tmp/pmcb_$i.log
done
python - <<'PY'
import csv, collections, glob, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+'pmcb_*.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'conv_bwd_fused' not in k: continue
    print(k)
    for c,x in sorted(v.items()): print('    %-32s %14.0f' % (c, sum(x)/len(x)))
PY
