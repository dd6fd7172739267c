#!/usr/bin/env python3
"""profiles/r01/kernel_table.md: one row per kernel of the step -- launches per step and average duration from the
rocprofv3 trace of the graph replay (bench_graph_kernel_stats.csv), HBM traffic per launch from the PMC passes
(pmc_summary.json), the resulting HBM rate, and each kernel's share of the step."""
import csv
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r01")
rows = [r for r in csv.DictReader(open(os.path.join(prof, "bench_graph_kernel_stats.csv")))]
pmc = json.load(open(os.path.join(prof, "pmc_summary.json")))
bench = json.load(open(os.path.join(prof, "bench_n1.json")))


def short(n):
    n = n.replace("void ", "").replace("sn::", "")
    return n[:n.index("(")] if "(" in n else n


steps = None
for r in rows:
    if "pairscan_kernel" in r["Name"]:
        pass
mine = [r for r in rows if "sn::" in r["Name"] or r["Name"].startswith(("void chamfer", "sigma_grad", "step_loss", "void sn"))]
# launches per step: the conv5 backward runs once per step (+ the roofline timing loop of bench.py: 50 launches + warm-up)
ref = [r for r in mine if "conv_bwd_bx3_kernel<64, 128" in r["Name"] or "conv_bwd_fused_kernel<64, 128" in r["Name"]][0]
steps = int(ref["Calls"])
tot = 0.0
lines = []
for r in mine:
    per_step = int(r["Calls"]) / steps
    if per_step < 0.5:  # (warm-up / probe launches that are not part of the step)
        continue
    avg = float(r["AverageNs"]) / 1e3
    t = pmc.get(r["Name"], {}).get("hbm_traffic_bytes_per_launch")
    n = max(1, round(per_step))
    in_step = avg * n
    tot += in_step
    lines.append((in_step, short(r["Name"]), n, avg, t))
lines.sort(reverse=True)
with open(os.path.join(prof, "kernel_table.md"), "w") as f:
    f.write("# Kernels of one sampler training step (B = 32, 1024 -> 64, K = 8), graph replay on 1 x MI355X\n\n")
    f.write("bench (THIS profile's box, bench_n1.json beside this file; the driver's BENCH_rNN.json comes from another box of the pool: "
            "+- 3 %%): %.1f k clouds/s, %.1f us/step; summed kernel time below: %.1f us (the rest is inter-kernel gap).\n" %
            (bench["value"] / 1e3, bench["ms_per_step"] * 1e3, tot))
    f.write("HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes; rate = traffic / duration.\n\n")
    f.write("| kernel | launches/step | avg us | share | HBM MB/launch | HBM GB/s |\n|---|---|---|---|---|---|\n")
    for in_step, name, n, avg, t in lines:
        f.write("| `%s` | %d | %.1f | %.1f %% | %s | %s |\n" % (name, n, avg, 100 * in_step / tot,
                                                          "%.2f" % (t / 1e6) if t else "-", "%.0f" % (t / avg / 1e3) if t else "-"))
print(open(os.path.join(prof, "kernel_table.md")).read())
