#!/usr/bin/env python3
"""Host-side profile of the eager module-surface step (what an unmodified train script issues): wall time per step, cProfile of
the calling thread, and wall-clock timers around the custom autograd nodes' forward / backward (the backward runs on autograd's
device thread, which cProfile does not see)."""
import cProfile
import collections
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd import SampleNet, ops, pointnet  # noqa: E402

with_task = len(sys.argv) > 1 and sys.argv[1] == "task"
with_sink = len(sys.argv) > 1 and sys.argv[1] == "sink"
torch.manual_seed(0)
B, N, M, K = 32, 1024, 64, 8
net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").cuda().train()
x = torch.rand(B, N, 3, device="cuda") - 0.5
if with_task:
    from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss

    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").cuda().eval()
    for p in pcr.parameters():
        p.requires_grad_(False)
    template = torch.rand(B, N, 3, device="cuda") - 0.5

acc = collections.defaultdict(float)
if with_sink:
    from samplenet_amd.parallel import FlatGradAllReducer

    red = FlatGradAllReducer(net)


def timed(owner, name, label):
    fn = getattr(owner, name)
    raw = fn.__func__ if hasattr(fn, "__func__") else fn

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0

    setattr(owner, name, staticmethod(wrapper) if isinstance(owner, type) else wrapper)


timed(pointnet, "forward_impl", "mlp forward_impl")
timed(pointnet, "backward_impl", "mlp backward_impl")
for cls in (pointnet.PointNetMLPFunction, ops.SoftProjectFunction, ops.SimplificationLossFunction, ops.ChamferFromScanFunction,
            ops.SamplerLossFunction, ops.ChamferDistanceFunction):
    timed(cls, "forward", cls.__name__ + ".forward")
    timed(cls, "backward", cls.__name__ + ".backward")


def step():
    if with_sink:
        red.zero_grad()
    else:
        for p in net.parameters():
            p.grad = None
    t0 = time.perf_counter()
    simp, proj = net(x)
    t1 = time.perf_counter()
    task = pcrnet_chamfer_loss(pcr, template, proj)[0] if with_task else proj.mean()
    loss = 0.01 * net.get_simplification_loss(x, simp, M, 1, 0) + 0.01 * net.get_projection_loss() + task
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    acc["net(x)"] += t1 - t0
    acc["losses"] += t2 - t1
    acc["loss.backward()"] += t3 - t2
    return loss


for plan in (True, False, True, False):
    pointnet.FORWARD_PLAN = plan
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    print("eager %s, forward plan %s: %.3f ms/step" % ("with PCRNet task" if with_task else "mean(proj)", plan, (time.perf_counter() - t0) / 300 * 1e3))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:7]:
        print("   %-40s %.1f us/step" % (k, v / 300 * 1e6))
if len(sys.argv) > 2:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
