#!/usr/bin/env python3
"""Phase stamps of skinny_linear_kernel on PCRNet's trunk layers (timeline build, SAMPLENET_AMD_LIB=tools/_ab/libsamplenet_hip_tl.so):
thread 0 of every workgroup, 100 MHz clock: start, operands loaded + MFMAs done, waves' partials summed, slice partial stored and
drained, arrival passed, (last workgroup of a tile) slices summed."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd._lib import check, lib, ptr  # noqa: E402

vp = ctypes.c_void_p
lib.sn_debug_timeline.argtypes = [vp, ctypes.c_int, ctypes.c_int]
R = 32
for K, N, tr in ((2048, 1024, 0), (1024, 1024, 0), (1024, 512, 0), (512, 256, 0), (256, 7, 0), (1024, 2048, 1)):
    x = torch.randn(R, K, device="cuda")
    W = torch.randn((K, N) if tr else (N, K), device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(R, N, device="cuda")
    scratch = torch.empty(lib.sn_skinny_linear_scratch_bytes(R, K, N) // 4, device="cuda")
    counters = torch.zeros(64, device="cuda", dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    big = torch.empty(64 << 20, device="cuda")  # (evicts the weights from the L2s between launches, as a training step does)
    for i in range(4):
        big.zero_()
        if i == 3:
            torch.cuda.synchronize()
            assert lib.sn_debug_timeline(None, 0, 1) == 0
        check(lib.sn_skinny_linear(R, K, N, ptr(x), None, ptr(W), tr, ptr(b), 1, ptr(out), ptr(scratch), ptr(counters), st), "skinny")
    torch.cuda.synchronize()
    tiles = (N + 31) // 32
    nb = 1024
    host = np.zeros((nb, 16), dtype=np.uint64)
    assert lib.sn_debug_timeline(host.ctypes.data_as(vp), nb, 0) == 0
    valid = host[:, 0] > 0
    t = host.astype(np.float64) / 100.0
    t0 = t[valid, 0].min()
    names = ["start", "loads + MFMAs done", "waves summed", "partial stored + drained", "arrival passed", "slices summed (last)"]
    print("K=%d N=%d %s: %d workgroups, %d tiles; last stamp at %.2f us" % (K, N, "transposed" if tr else "", valid.sum(), tiles,
                                                                          (t[valid][:, :6].max() - t0)))
    prev = None
    for k, nm in enumerate(names):
        ok = valid & (host[:, k] > 0)
        if not ok.any():
            continue
        col = t[ok, k] - t0
        print("   %-26s median at %6.2f us (p10 %6.2f  p90 %6.2f)  n=%d" % (nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90), ok.sum()))
