#!/usr/bin/env python3
"""sn_linear_forward_maxpool_wide alone (HIP events, back-to-back launches) on the registration step's two shapes:
template 32 x 1024 points and sampled cloud 32 x 64 points, 128 -> 1024 channels.  SAMPLENET_AMD_LIB selects a variant build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd._lib import check, lib, ptr  # noqa: E402

torch.manual_seed(0)
Ci, Co = 128, 1024
for B, N in ((32, 1024), (16, 1024)):
    R = B * N
    a = torch.randn(R, Ci, device="cuda")
    coef = torch.zeros(4, Ci, device="cuda")
    coef[0] = 1
    W = torch.randn(Co, Ci, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda")
    pooled = torch.empty(B, Co, device="cuda")
    argsel = torch.empty(B, Co, device="cuda", dtype=torch.int32)
    zsel = torch.empty(B, Co, device="cuda")
    planes = torch.empty(3 * Co * Ci, device="cuda", dtype=torch.bfloat16)
    scratch = torch.empty(lib.sn_linear_forward_maxpool_wide_scratch_bytes(R, Ci, Co, N) // 8, device="cuda", dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream

    def run(ready):
        # B = 32: no gradient, no argmax; B = 16: rows of the maxima for a backward
        check(lib.sn_linear_forward_maxpool_wide(R, Ci, Co, N, ptr(a), ptr(coef), ptr(W), ptr(b), None, ptr(scratch), ptr(pooled),
                                                 ptr(argsel) if B == 16 else None, ptr(zsel) if B == 16 else None, ptr(planes),
                                                 ready, st), "wide")

    run(0)
    for _ in range(5):
        run(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        run(1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("B=%d N=%d: %.1f us per call (GEMM + decode)  %.1f fp32-equivalent TFLOP/s" % (B, N, us, 2.0 * R * Ci * Co / us * 1e-6))
