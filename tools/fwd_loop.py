#!/usr/bin/env python3
"""Runs the production forward of the sampler's MLP (conv stack + FC head: pointnet.forward_impl) on the headline shape, for
rocprofv3 --pmc passes over the forward GEMM kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd import SampleNet, pointnet  # noqa: E402

torch.manual_seed(0)
net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
B = int(os.environ.get("SQ_BATCH", "32"))  # (2048: the saturating batch -- the persistent forward kernels)
x = (torch.rand(B, 1024, 3, device="cuda") - 0.5).contiguous()
with torch.no_grad():
    for _ in range(10 if B <= 64 else 4):
        pointnet.forward_impl(net, x, True)
torch.cuda.synchronize()
