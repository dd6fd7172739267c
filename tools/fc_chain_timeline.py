#!/usr/bin/env python3
"""Phase timeline of the two FC chain kernels inside the captured step (debug build: tools/build_variant.sh tl fc_chain.hip
-DSN_TIMELINE; run with SAMPLENET_AMD_LIB=tools/_ab/libsamplenet_hip_tl.so).  Thread 0 of every chain workgroup stamps the
100 MHz wall clock at the phase boundaries; printed: per phase the median over the workgroups of (stamp - kernel's first stamp)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd import SampleNet  # noqa: E402
from samplenet_amd._lib import LIB_PATH  # noqa: E402
from samplenet_amd.engine import SamplerTrainStep  # noqa: E402
from samplenet_amd.parallel import FlatGradAllReducer  # noqa: E402

dbg = ctypes.CDLL(LIB_PATH)
torch.manual_seed(0)
net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
x = torch.rand(32, 1024, 3, device="cuda") - 0.5
step = SamplerTrainStep(net, x, reducer=FlatGradAllReducer(net), use_graph=True)
for _ in range(20):
    step(x)
torch.cuda.synchronize()
host = np.zeros((2, 16, 32), dtype=np.uint64)
assert dbg.sn_debug_fc_timeline(host.ctypes.data_as(ctypes.c_void_p)) == 0
t = host.astype(np.float64) / 100.0  # us
names_f = {0: "start", 1: "pool stage + weights staged", 24: "  (sums landed + totalled)", 25: "  (coefficients in LDS)", 26: "  (keys decoded, pooled in LDS)", 27: "  (weight slices in LDS)"}
for l in range(3):
    names_f.update({2 + 6 * l: "L%d MFMA + wave sum" % l, 3 + 6 * l: "L%d epilogue" % l, 4 + 6 * l: "L%d tile stored + drained" % l,
                    5 + 6 * l: "L%d seam passed" % l, 6 + 6 * l: "L%d gather landed" % l, 7 + 6 * l: "L%d staged" % l})
names_f[31] = "end (drained)"
names_b = {0: "start (epoch read)", 1: "stage-0 operands staged"}
for s in range(4):
    names_b.update({2 + 6 * s: "S%d MFMA + wave sum" % s, 3 + 6 * s: "S%d epilogue" % s, 4 + 6 * s: "S%d tile stored + drained" % s,
                    5 + 6 * s: "S%d arrivals passed" % s, 6 + 6 * s: "S%d gather landed" % s, 7 + 6 * s: "S%d staged" % s})
names_b[31] = "end (drained)"
for kind, label, names, wgs in ((0, "fc_chain_fwd_kernel", names_f, range(8)), (1, "fc_chain_bwd_kernel: chain workgroups", names_b, range(8)),
                                (1, "fc_chain_bwd_kernel: weight-gradient workgroups", {0: "start", 31: "end", 2: "S0 done", 8: "S1 done", 14: "S2 done", 20: "S3 done"}, range(8, 16))):
    tt = t[kind][list(wgs)]
    t0 = tt[:, 0][tt[:, 0] > 0].min()
    print("== %s" % label)
    prev = 0.0
    for k in sorted(names, key=lambda k: (k if k < 24 or k == 31 else 0.5 + k / 100.0)):
        col = tt[:, k]
        ok = col > 0
        if not ok.any():
            continue
        med = float(np.median(col[ok] - t0))
        print("   %-32s at %6.2f us  (+%5.2f)   min %6.2f max %6.2f" % (names[k], med, med - prev, float((col[ok] - t0).min()), float((col[ok] - t0).max())))
        prev = med
