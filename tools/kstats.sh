# per-kernel average durations of the step at batch B for library variants: bash tools/kstats.sh B NAME1 NAME2 ...   ("main" = product build)
cd /tmp && export TMPDIR=/tmp
B=$1; shift
for v in "$@"; do
  if [ "$v" == "main" ]; then unset SAMPLENET_AMD_LIB; else export SAMPLENET_AMD_LIB=$GRAFT_REPO_ROOT/tools/_ab/libsamplenet_hip_$v.so; fi
  rm -rf /tmp/ks_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps ${STEPS:-60} --warmup 10 --no-probes --min-time 0 > /tmp/ks_$v.log 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ks_$v/k_kernel_stats.csv')))
tot=0
print('== $v B=$B')
for r in rows:
    n=r['Name']
    if 'sn::' in n or 'chamfer' in n or 'step_loss' in n or 'sigma' in n:
        if int(r['Calls'])>=10:
            tot+=float(r['AverageNs'])*int(r['Calls'])/ (int(rows[0]['Calls']) if False else 1)
            print('  %-95s %5s %9.2f us'%(n[:95], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
