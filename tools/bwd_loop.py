#!/usr/bin/env python3
"""Runs sn_linear_backward (production library) on the three fused-backward shapes, for rocprofv3 --pmc passes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd._lib import lib, ptr  # noqa: E402


def bwd(R, Ci, Co, mode, B=32, reps=10):
    dev = "cuda"
    npts = R // B
    dy = torch.randn(R, Co, device=dev) if mode != 2 else None
    z = torch.randn(R, Co, device=dev)
    kc = torch.randn(3, Co, device=dev)
    gsel = torch.randn(B, Co, device=dev) if mode == 2 else None
    argsel = torch.randint(0, npts, (B, Co), device=dev, dtype=torch.int32) if mode == 2 else None
    W = torch.randn(Co, Ci, device=dev) * 0.1
    zprev = torch.randn(R, Ci, device=dev)
    coefp = torch.rand(4, Ci, device=dev) + 0.5
    dyprev = torch.empty(R, Ci, device=dev)
    stats = torch.empty(lib.sn_linear_stats_blocks(R), 2, Ci, device=dev)
    ns = lib.sn_linear_wgrad_splits(R, Ci, Co, 0)
    part = torch.empty(ns * Co * Ci, device=dev)
    dW = torch.empty(Co, Ci, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(reps):
        rc = lib.sn_linear_backward(R, Ci, Co, mode, ptr(dy), ptr(z), ptr(kc), ptr(gsel), ptr(argsel), npts, ptr(W), ptr(zprev),
                                    ptr(coefp), ptr(dyprev), ptr(stats), ptr(part), ptr(dW), st)
        assert rc == 0
    torch.cuda.synchronize()


if __name__ == "__main__":
    B = int(os.environ.get("SQ_BATCH", "32"))  # (2048: the saturating batch)
    R = B * 1024
    reps = 10 if B <= 64 else 4
    bwd(R, 64, 64, 1, B=B, reps=reps)
    bwd(R, 64, 128, 1, B=B, reps=reps)
    bwd(R, 128, 128, 2, B=B, reps=reps)
