#!/usr/bin/env python3
"""gpurun_out/r01/pmc_{FETCH_SIZE,WRITE_SIZE}.csv (rocprofv3 --pmc, one counter per run) -> profiles/r01/pmc_summary.json:
per kernel the mean counter values and HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB -- on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r01")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    if "sn::" not in k and not k.startswith(("simp", "sigma", "step_loss", "sampler_loss", "void sn", "void chamfer")):
        continue
    d = {c: sum(x) / len(x) for c, x in v.items()}
    d["launches"] = max(len(x) for x in v.values())
    d["hbm_traffic_bytes_per_launch"] = (2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024
    out[k] = {kk: round(vv, 1) for kk, vv in d.items()}
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", "r01", "pmc_summary.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in sorted(out, key=lambda n: -out[n]["hbm_traffic_bytes_per_launch"])[:12]:
    print("%-90s %8.2f MB/launch" % (k[:90], out[k]["hbm_traffic_bytes_per_launch"] / 1e6))
