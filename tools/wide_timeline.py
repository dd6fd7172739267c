#!/usr/bin/env python3
"""Phase stamps of linear_fwd_wide_pool_kernel (timeline build: tools/build_timeline_lib.sh; run with
SAMPLENET_AMD_LIB=tools/_ab/libsamplenet_hip_tl.so): start, prologue done, iteration 4: top / MFMAs done / next block staged /
barrier passed, end -- thread 0 (wave 0) and thread 256 (wave 4) of every workgroup, 100 MHz clock."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd._lib import check, lib, ptr  # noqa: E402

vp = ctypes.c_void_p
B, N, Ci, Co = 32, 1024, 128, 1024
R = B * N
a = torch.randn(R, Ci, device="cuda")
coef = torch.zeros(4, Ci, device="cuda")
coef[0] = 1
W = torch.randn(Co, Ci, device="cuda") * 0.1
b = torch.randn(Co, device="cuda")
pooled = torch.empty(B, Co, device="cuda")
planes = torch.empty(3 * Co * Ci, device="cuda", dtype=torch.bfloat16)
scratch = torch.empty(lib.sn_linear_forward_maxpool_wide_scratch_bytes(R, Ci, Co, N) // 8, device="cuda", dtype=torch.int64)
st = torch.cuda.current_stream().cuda_stream
for i in range(4):
    if i == 3:
        torch.cuda.synchronize()
        lib.sn_debug_timeline.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        assert lib.sn_debug_timeline(None, 0, 1) == 0
    check(lib.sn_linear_forward_maxpool_wide(R, Ci, Co, N, ptr(a), ptr(coef), ptr(W), ptr(b), None, ptr(scratch), ptr(pooled), None, None,
                                             ptr(planes), int(i > 0), st), "wide")
torch.cuda.synchronize()
nb = R // 128
host = np.zeros((nb, 16), dtype=np.uint64)
assert lib.sn_debug_timeline(host.ctypes.data_as(vp), nb, 0) == 0
t = host.astype(np.float64) / 100.0
t0 = t[:, 0].min()
names = ["start", "prologue done", "it4 top", "it4 MFMAs + fillers done", "it4 next block staged", "it4 barrier passed", "end"]
for w, lab in ((0, "wave 0"), (8, "wave 4")):
    print("[%s]" % lab)
    prev = None
    for k, nm in enumerate(names):
        col = t[:, w + k] - t0
        d = col - prev if prev is not None else col
        print("   %-28s +%6.2f us (p10 %6.2f  p90 %6.2f)   at %6.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(col)))
        prev = col
