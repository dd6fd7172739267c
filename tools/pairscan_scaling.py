#!/usr/bin/env python3
"""Pair-scan kernel (kNN + soft projection + both Chamfer directions) at growing batch: where the B = 32 launch sits on
the kernel's own throughput curve.  Prints clouds/s and algorithmic GB/s (24,576 B per cloud at N=1024, M=64, K=8).
    python tools/pairscan_scaling.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd._lib import check, lib, ptr  # noqa: E402

N, M, K = 1024, 64, 8
for B in ([int(v) for v in sys.argv[1:]] or [32, 128, 512, 2048, 8192]):
    dev = "cuda"
    x = torch.rand(B, N, 3, device=dev) - 0.5
    y = torch.rand(B, 3, M, device=dev) - 0.5
    T = torch.ones(1, device=dev)
    proj = torch.empty(B, M, 3, device=dev)
    idx = torch.empty(B, M, K, device=dev, dtype=torch.int32)
    dq, iq = torch.empty(B, M, device=dev), torch.empty(B, M, device=dev, dtype=torch.int32)
    dp, ip = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev, dtype=torch.int32)
    wsb = lib.sn_pairscan_workspace_bytes(B, N, M)
    ws = torch.empty(max(wsb // 8, 1), device=dev, dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        check(lib.sn_pairscan_forward_ws(B, N, M, K, ptr(x), 0, ptr(y), 1, ptr(idx), None, ptr(dq), ptr(iq), ptr(dp), ptr(ip),
                                         ptr(proj), 0, None, ptr(T), 1e-2, ptr(ws) if wsb else None, wsb, st), "pairscan")

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = (12 * N + 12 * M + 4 * M * K + 12 * M + 8 * M + 8 * N) * B
    print("B=%5d: %8.1f us per launch(es)  %9.0f clouds/s  %7.1f GB/s algorithmic  (%.3f us per cloud)" %
          (B, ms * 1e3, B / (ms * 1e-3), alg / (ms * 1e-3) / 1e9, ms * 1e3 / B))
