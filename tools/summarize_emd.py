#!/usr/bin/env python3
"""EMD profile passes (tools/gpu_profile.sh: rocprofv3 --pmc over tools/emd_loop.py, one form per run) ->
  emd_pmc_summary.json  {form: HBM bytes per call}   = sum over the form's kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB / reps
  emd_sq_counters.json  {form: VALU-busy fraction}   = sum SQ_ACTIVE_INST_VALU / sum SQ_BUSY_CYCLES-equivalent wave time, plus the raw sums
usage: summarize_emd.py SRC_DIR DST_DIR REPS"""
import collections
import csv
import json
import os
import sys

src, dst, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
pmc, sq = {}, {}
for form in ("emd_loss", "three_call"):
    tot = collections.defaultdict(float)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(src, "emd_%s_pmc_%s.csv" % (form, c))
        if os.path.exists(path):
            for r in csv.DictReader(open(path)):
                if "sn::" in r["Kernel_Name"] or "emd" in r["Kernel_Name"].lower() or "match" in r["Kernel_Name"].lower():
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
    if tot:
        pmc[form] = (2 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024 / reps
    path = os.path.join(src, "emd_%s_sq.csv" % form)
    if os.path.exists(path):
        s = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            s[r["Counter_Name"]] += float(r["Counter_Value"])
        d = {k: v for k, v in s.items()}
        # SQ_* wave counters are quad-cycles summed over waves; SQ_BUSY_CYCLES per SQ instance (32 shader engines' worth).
        # VALU busy = cycles a SIMD's VALU issued / cycles the kernel ran: SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)
        if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
            d["valu_busy"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
        if d.get("SQ_WAVE_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
            d["valu_share_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
        sq[form] = d
json.dump(pmc, open(os.path.join(dst, "emd_pmc_summary.json"), "w"), indent=1)
json.dump({k: {"valu_busy": v.get("valu_busy"), "valu_share_of_wave_cycles": v.get("valu_share_of_wave_cycles"),
               "counters": {c: x for c, x in v.items() if c.startswith("SQ_")},
               "note": "rocprofv3 --pmc over tools/emd_loop.py (B=50, 2048x2048), sums over the form's kernels; SQ wave counters in quad-cycles"}
           for k, v in sq.items()}, open(os.path.join(dst, "emd_sq_counters.json"), "w"), indent=1)
print(json.dumps(pmc), json.dumps({k: (v.get("valu_busy"), v.get("valu_share_of_wave_cycles")) for k, v in sq.items()}))
