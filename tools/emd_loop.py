#!/usr/bin/env python3
"""One form of the EMD loss at BASELINE configs[3] (B = 50, n = m = 2048) in a short loop, for rocprofv3 passes:
    python tools/emd_loop.py emd_loss | three_call  [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd import ops  # noqa: E402

form, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, n, m = 50, 2048, 2048
g = torch.Generator(device="cuda").manual_seed(3)
a = (torch.rand(B, n, 3, device="cuda", generator=g) - 0.5).requires_grad_(True)
b = (torch.rand(B, m, 3, device="cuda", generator=g) - 0.5).requires_grad_(True)
for _ in range(reps):
    if form == "emd_loss":
        torch.autograd.grad(ops.emd_loss(a, b).sum(), [a, b])
    else:
        torch.autograd.grad(ops.match_cost(a, b, ops.approx_match(a, b)).sum(), [a, b])
torch.cuda.synchronize()
