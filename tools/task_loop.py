#!/usr/bin/env python3
"""The registration task term alone (frozen PCRNet on (template 1024 pts, 64 projected pts) + Chamfer, forward + gradient to the
projected points: registration/main.py:557-577) in a loop -- for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss  # noqa: E402

B, N, M = 32, 1024, 64
torch.manual_seed(0)
pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").cuda().eval()
for p in pcr.parameters():
    p.requires_grad_(False)
template = torch.rand(B, N, 3, device="cuda") - 0.5
q = (torch.rand(B, M, 3, device="cuda") - 0.5).requires_grad_(True)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    q.grad = None
    loss = pcrnet_chamfer_loss(pcr, template, q)[0]
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
