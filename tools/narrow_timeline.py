#!/usr/bin/env python3
"""Phase stamps of pointnet_narrow_fwd_kernel (timeline build, SAMPLENET_AMD_LIB=tools/_ab/libsamplenet_hip_tl.so): thread 0 of
every workgroup, 100 MHz clock."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd._lib import check, lib, ptr  # noqa: E402

vp = ctypes.c_void_p
lib.sn_debug_timeline.argtypes = [vp, ctypes.c_int, ctypes.c_int]
for B, N in ((32, 1024), (32, 64)):
    R = B * N
    x = torch.rand(R, 3, device="cuda") - 0.5
    Ws = [torch.randn(64, 3, device="cuda"), torch.randn(64, 64, device="cuda") * 0.1, torch.randn(64, 64, device="cuda") * 0.1,
          torch.randn(128, 64, device="cuda") * 0.1]
    bs = [torch.randn(w.shape[0], device="cuda") for w in Ws]
    planes = torch.empty(3 * 16384, device="cuda", dtype=torch.bfloat16)
    z4 = torch.empty(R, 128, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(4):
        if i == 3:
            torch.cuda.synchronize()
            assert lib.sn_debug_timeline(None, 0, 1) == 0
        check(lib.sn_pointnet_narrow_forward(R, ptr(x), ptr(Ws[0]), ptr(bs[0]), ptr(Ws[1]), ptr(bs[1]), ptr(Ws[2]), ptr(bs[2]), ptr(Ws[3]),
                                             ptr(bs[3]), ptr(planes), int(i > 0), None, None, None, ptr(z4), st), "narrow")
    torch.cuda.synchronize()
    nb = R // 128
    host = np.zeros((nb, 16), dtype=np.uint64)
    assert lib.sn_debug_timeline(host.ctypes.data_as(vp), nb, 0) == 0
    t = host.astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    names = ["start", "planes staged", "conv1 -> fragments", "conv2 + transpose", "conv3 + transpose", "conv4 MFMAs", "end (stores issued)"]
    print("B=%d N=%d (%d workgroups)" % (B, N, nb))
    prev = None
    for k, nm in enumerate(names):
        col = t[:, k] - t0
        d = col - prev if prev is not None else col
        print("   %-24s +%6.2f us (p10 %6.2f  p90 %6.2f)   at %6.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(col)))
        prev = col
