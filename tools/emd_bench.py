#!/usr/bin/env python3
"""EMD row (a11) on the GPU: sn_approxmatch / sn_matchcost / sn_matchcost_grad at the reconstruction config's size
(BASELINE configs[3]: 2048-point clouds) and at the classification size, timed with HIP events; checks the match against
the oracle on one small case first."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


for (B, n, m) in [(32, 1024, 64), (8, 2048, 2048), (50, 2048, 2048)]:
    a = torch.rand(B, n, 3, device="cuda") - 0.5
    b = torch.rand(B, m, 3, device="cuda") - 0.5
    t_match, match = timed(lambda: ops.approx_match(a, b), 5)
    a.requires_grad_(True), b.requires_grad_(True)
    t_cost, cost = timed(lambda: ops.match_cost(a, b, match), 5)
    t_grad, _ = timed(lambda: torch.autograd.grad(ops.match_cost(a, b, match).sum(), [a, b]), 5)
    mb = B * n * m * 4 / 1e6
    print("B=%3d n=%4d m=%4d: approxmatch %8.1f us (%6.1f us/cloud, match %7.1f MB -> %6.1f GB/s written once)  matchcost %7.1f us  "
          "cost+grad %7.1f us   row sums of match: %.4f..%.4f" %
          (B, n, m, t_match * 1e3, t_match * 1e3 / B, mb, mb / t_match, t_cost * 1e3, t_grad * 1e3,
           float(match.sum(2).min()), float(match.sum(2).max())))

# the loss without the match matrix (sn_emd_loss: auction + two sweeps) against the three-call composition
for (B, n, m) in [(50, 2048, 2048)]:
    a = (torch.rand(B, n, 3, device="cuda") - 0.5).requires_grad_(True)
    b = (torch.rand(B, m, 3, device="cuda") - 0.5).requires_grad_(True)
    t_fused, _ = timed(lambda: torch.autograd.grad(ops.emd_loss(a, b).sum(), [a, b]), 5)
    t_three, _ = timed(lambda: torch.autograd.grad(ops.match_cost(a, b, ops.approx_match(a, b)).sum(), [a, b]), 5)
    print("B=%3d n=%4d m=%4d: emd loss + gradients: no-materialise %8.1f us, approx_match + match_cost + grad %8.1f us" %
          (B, n, m, t_fused * 1e3, t_three * 1e3))
