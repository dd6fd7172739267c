"""Host-side profile of the reference call pattern on the captured surface: python tools/surface_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from samplenet_amd import SampleNet  # noqa: E402

B, N, M, K = 32, 1024, 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
x = torch.rand(B, N, 3, device=dev) - 0.5
params = list(net.parameters())


def step(t=None):
    t0 = time.perf_counter()
    for p in params:
        p.grad = None
    t1 = time.perf_counter()
    simp, proj = net(x)
    t2 = time.perf_counter()
    ls = net.get_simplification_loss(x, simp, M, 1, 0)
    lp = net.get_projection_loss()
    t3 = time.perf_counter()
    loss = 0.01 * ls + 0.01 * lp + proj.mean()
    t4 = time.perf_counter()
    loss.backward()
    t5 = time.perf_counter()
    if t is not None:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            t[i] += d
    return loss


for _ in range(10):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
t = [0.0] * 5
t0 = time.perf_counter()
for _ in range(n):
    step(t)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("host %.1f us/step, wall %.1f us/step" % (host / n * 1e6, wall / n * 1e6))
print("grad=None %.1f | net(x) %.1f | getters %.1f | loss arithmetic %.1f | backward %.1f  (us/step)" % tuple(v / n * 1e6 for v in t))
# with a device synchronisation after every step: pure GPU time of a step + launch latency
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
    torch.cuda.synchronize()
print("synchronised per step: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

# ---- where the backward's time goes (it runs on autograd's device thread: invisible to cProfile above) -------------------
from samplenet_amd import surface  # noqa: E402

acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r

    return w


plan = surface.plans(net)[0]
plan.commit_begin = timed("commit_begin", plan.commit_begin)
plan.commit_end = timed("commit_end", plan.commit_end)
plan.gb.replay = timed("gb.replay", plan.gb.replay)
plan.gf.replay = timed("gf.replay", plan.gf.replay)
orig_bwd = surface._SurfaceFunction.backward
surface._SurfaceFunction.backward = staticmethod(timed("node.backward", orig_bwd))
orig_fwd = surface._SurfaceFunction.forward
surface._SurfaceFunction.forward = staticmethod(timed("node.forward", orig_fwd))
surface.try_forward = timed("try_forward", surface.try_forward)
import samplenet_amd.samplenet as _sm  # noqa: E402

for _ in range(n):
    step()
torch.cuda.synchronize()
print({k: round(v / n * 1e6, 1) for k, v in acc.items()}, "us/step")
