#!/usr/bin/env python3
"""gpurun_out/pmcb_*.csv (tools/gpu_pmc_bwd.sh: three rocprofv3 --pmc passes of 8 SQ counters over tools/bwd_loop.py) ->
profiles/r01/conv_bwd_fused_sq_counters.json: mean counter values per fused-backward kernel and the derived MFMA utilisation
(SQ_VALU_MFMA_BUSY_CYCLES per SIMD over the kernel's cycles) and LDS bank-conflict share."""
import collections
import csv
import glob
import json
import os

import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: summarize_sq.py [csv-glob] [kernel-name-substring] [output json]
pattern = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmcb_*.csv")
needle = sys.argv[2] if len(sys.argv) > 2 else "conv_bwd_fused"
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles", "r01", "conv_bwd_fused_sq_counters.json")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(pattern):
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = sys.argv[4] if len(sys.argv) > 4 else "32768"
out = {"note": "rocprofv3 --pmc, passes of 8 SQ counters each (tools/gpu_sq_r06.sh / gpu_profile_sq.sh), means over the launches of the profiled loop at R = " + rows + "; "
               "SQ_* wave counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over all SIMDs", "kernels": {}}
for k, v in sorted(agg.items()):
    d = {c: int(round(sum(x) / len(x))) for c, x in sorted(v.items())}
    der = {}
    if "SQ_BUSY_CYCLES" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        cyc = d["SQ_BUSY_CYCLES"] / 32.0            # 32 shader engines' worth of SQ instances
        mfma = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0  # 256 CUs x 4 SIMDs
        der["kernel_cycles_per_SE(SQ_BUSY_CYCLES/32)"] = int(round(cyc))
        der["mfma_busy_cycles_per_SIMD(SQ_VALU_MFMA_BUSY_CYCLES/1024)"] = int(round(mfma))
        der["mfma_utilisation"] = round(mfma / cyc, 3)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_fraction_of_lds_active"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 3)
    d["derived"] = der
    out["kernels"][k] = d
    print(k[:80], der)
json.dump(out, open(dst, "w"), indent=1)
