python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|Error|^FAILED" | head -40
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('graph', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-graph > /tmp/bench_prof.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_r1d; cp /tmp/prof/bench_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_r1d/
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/prof/bench_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel us per step", tot/1e3/123)
for r in rows[:24]:
    print("%-86s calls/step %5.1f avg %8.1f us %5.2f%%" % (r['Name'][:86], int(r['Calls'])/123, float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
