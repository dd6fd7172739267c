# LDS map of every kernel of the headline step, the B = 2048 step and the registration task term: SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT /
# SQ_BUSY_CYCLES per launch (one rocprofv3 --pmc pass each) -> gpurun_out/<round>/lds_map.txt.  Run: gpurun --timeout 600 -- 'bash tools/gpu_lds_map.sh r06'
R=${1:-r06}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/lm_b32 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-probes --no-cpu-baseline --no-graph > /tmp/lm1.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/lm_b2048 -o p -- python $GRAFT_REPO_ROOT/bench.py --batch 2048 --steps 4 --warmup 2 --no-probes --no-cpu-baseline --no-graph > /tmp/lm2.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/lm_task -o p -- python $GRAFT_REPO_ROOT/tools/task_loop.py 10 > /tmp/lm3.log 2>&1
python - <<'PY' > $OUT/lds_map.txt
import csv, collections
for tag in ("b32", "b2048", "task"):
    try:
        rows = list(csv.DictReader(open("/tmp/lm_%s/p_counter_collection.csv" % tag)))
    except Exception as e:
        print(tag, "missing", e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s: per launch -- LDS-active cycles (sum over CUs), bank-conflict share of them, LDS-active / (SQ_BUSY_CYCLES x 8: CU-cycles), wave time waiting on LDS" % tag)
    out = []
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        act, conf, busy = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_BUSY_CYCLES", 0)
        if act < 1000: continue
        out.append((act, "%-100s act %10.0f  conflict %4.0f %%  lds-busy %4.1f %%  wait-lds %4.1f %%" % (
            k[:100], act, 100 * conf / max(act, 1), 100 * act / max(busy * 8, 1), 100 * m.get("SQ_WAIT_INST_LDS", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1))))
    for _, l in sorted(out, reverse=True)[:24]: print(l)
PY
cat $OUT/lds_map.txt | cut -c1-200
