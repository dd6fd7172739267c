"""Multi-process (world_size = 2, gloo, CPU) test of the data-parallel layer: the batch is sharded, every rank runs
the sampler step on its shard, and the flat gradient bucket after FlatGradAllReducer.reduce() equals the gradient of a
single process on the whole batch.  The network is the CPU restatement of the reference module
(oracle/cpu_reference_model.py) in eval-mode BatchNorm (running statistics), which makes the equality exact up to
fp32 summation order; the reducer code under test is the one bench.py runs on RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss(net, x):
    simp, proj = net(x)
    return 0.01 * net.get_simplification_loss(x, simp, 16, 1, 0) + 0.01 * net.sigma() + proj.mean()


def _worker(rank, world, port, x_all, state, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.cpu_reference_model import SampleNetCPU
    from samplenet_amd.parallel import FlatGradAllReducer, shard_batch

    torch.set_num_threads(1)
    net = SampleNetCPU(16, 32, 4)
    net.load_state_dict(state)
    net.eval()
    red = FlatGradAllReducer(net)
    assert red.world == world and not red.overlap
    x = shard_batch(x_all, rank, world)
    for _ in range(2):  # second iteration proves zero_grad() resets the bucket
        red.zero_grad()
        _loss(net, x).backward()
        red.reduce()
    torch.save(red.flat.clone(), os.path.join(out_dir, "flat%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_big_batch(tmp_path):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.cpu_reference_model import SampleNetCPU
    from samplenet_amd.parallel import FlatGradAllReducer, shard_batch

    torch.manual_seed(0)
    net = SampleNetCPU(16, 32, 4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x_all = torch.rand(8, 128, 3) - 0.5
    port = _free_port()
    mp.spawn(_worker, args=(2, port, x_all, state, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = torch.load(tmp_path / "flat0.pt"), torch.load(tmp_path / "flat1.pt")
    assert torch.equal(f0, f1)  # every rank holds the same averaged gradient
    net.eval()
    single = FlatGradAllReducer(net)  # world size 1: reduce() is a no-op, same flat layout
    single.zero_grad()
    _loss(net, x_all).backward()
    # mean of per-shard means == mean over the whole batch for every term but mean_b(max ...) -- also a batch mean
    assert torch.allclose(f0, single.flat, rtol=1e-4, atol=1e-7)
    with pytest.raises(ValueError):
        shard_batch(torch.zeros(7, 4, 3), 0, 2)
    assert torch.equal(shard_batch(x_all, 1, 2), x_all[4:])


def test_eight_rank_gradient_average_equals_big_batch(tmp_path):
    """The same equality at the world size the scaling target is quoted on (north_star: 8 ranks, global batch 8 x per-rank batch):
    eight gloo processes, two clouds each, the flat bucket of every rank after reduce() is bit-equal to every other rank's and
    matches the single-process gradient on all sixteen clouds (VERDICT r5 #7a; the reducer code is the one bench.py runs on RCCL)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.cpu_reference_model import SampleNetCPU
    from samplenet_amd.parallel import FlatGradAllReducer

    world = 8
    torch.manual_seed(1)
    net = SampleNetCPU(16, 32, 4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x_all = torch.rand(2 * world, 128, 3) - 0.5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, x_all, state, str(tmp_path)), nprocs=world, join=True)
    flats = [torch.load(tmp_path / ("flat%d.pt" % r)) for r in range(world)]
    for r in range(1, world):
        assert torch.equal(flats[0], flats[r]), r
    net.eval()
    single = FlatGradAllReducer(net)
    single.zero_grad()
    _loss(net, x_all).backward()
    assert torch.allclose(flats[0], single.flat, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------------------------------------
# The engine / reducer code path WITH the gradient-sink semantics (pointnet.GradSink: the first backward of a step overwrites the
# bucket views, further ones accumulate, views are re-bound after optimizer.zero_grad()) under two ranks.  The HIP kernels
# cannot run here, so `_SinkNet` stands in for the HIP model the way that matters to this code: its backward WRITES the
# gradients of `fc.*` into module._grad_sink's views itself and hands nothing to autograd; `project.t` arrives through
# autograd's accumulate like the temperature.
class _SinkLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, W, b):
        ctx.net = net
        ctx.save_for_backward(x, W)
        return x @ W.t() + b

    @staticmethod
    def backward(ctx, g):
        from samplenet_amd import pointnet

        x, W = ctx.saved_tensors
        fresh = {"fc.weight": g.t() @ x, "fc.bias": g.sum(0)}
        sink, owner = pointnet.sink_for_backward(ctx.net)
        if sink is not None:  # (the kernels' in-place route)
            for n, v in fresh.items():
                sink[n].copy_(v)
        if owner is not None:
            owner.commit(None if sink is not None else fresh)
            return None, None, None, None
        return None, None, fresh["fc.weight"], fresh["fc.bias"]


class _Project(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.t = torch.nn.Parameter(torch.tensor(1.5))


class _SinkNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(6, 4)
        self.project = _Project()

    def forward(self, x):
        return _SinkLinear.apply(self, x, self.fc.weight, self.fc.bias) * self.project.t


def _sink_loss(net, x):
    return (net(x) ** 2).mean()


def _sink_worker(rank, world, port, x_all, state, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from samplenet_amd.parallel import FlatGradAllReducer, shard_batch

    net = _SinkNet()
    net.load_state_dict(state)
    red = FlatGradAllReducer(net, kernel_written=["fc.weight", "fc.bias"])
    assert set(net._grad_sink) == {"fc.weight", "fc.bias"} and [p is net.project.t for p, _ in red._autograd] == [True]
    x = shard_batch(x_all, rank, world)
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    out = {}
    # 1. plain step: zero_grad() of the reducer, one backward (overwrites the views), reduce
    red.zero_grad()
    _sink_loss(net, x).backward()
    red.reduce()
    out["plain"] = red.flat.clone()
    # 2. optimizer.zero_grad() (set_to_none=True): the views are dropped; the backward's commit takes them back
    opt.zero_grad()
    assert net.fc.weight.grad is None and net.project.t.grad is None
    _sink_loss(net, x).backward()
    red.reduce()
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in
               [(net._grad_sink.params[n], v) for n, v in net._grad_sink.items()] + red._autograd)
    out["after_zero_grad"] = red.flat.clone()
    # 3. gradient accumulation over two micro-batches inside one step: the second backward must ADD
    red.zero_grad()
    h = x.shape[0] // 2
    _sink_loss(net, x[:h]).backward()
    _sink_loss(net, x[h:]).backward()
    red.reduce()
    out["accumulated"] = red.flat.clone()
    # 4. a graph REPLAY writes the views without any Python running: after opt.zero_grad() reduce() alone re-binds them
    opt.zero_grad()
    for n, v in net._grad_sink.items():
        v.fill_(float(rank + 1))
    for _, v in red._autograd:
        v.fill_(float(rank + 1))
        net.project.t.grad = v
    red.reduce()
    assert all(net._grad_sink.params[n].grad.data_ptr() == v.data_ptr() for n, v in net._grad_sink.items())
    out["replayed"] = red.flat.clone()
    opt.step()  # every parameter has a gradient again
    torch.save(out, os.path.join(out_dir, "sink%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_sink_semantics(tmp_path):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from samplenet_amd.parallel import FlatGradAllReducer

    torch.manual_seed(3)
    net = _SinkNet()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x_all = torch.randn(8, 6)
    mp.spawn(_sink_worker, args=(2, _free_port(), x_all, state, str(tmp_path)), nprocs=2, join=True)
    o0, o1 = torch.load(tmp_path / "sink0.pt"), torch.load(tmp_path / "sink1.pt")
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k
    single = FlatGradAllReducer(net, kernel_written=["fc.weight", "fc.bias"])
    single.zero_grad()
    _sink_loss(net, x_all).backward()
    assert torch.allclose(o0["plain"], single.flat, rtol=1e-5, atol=1e-7)
    assert torch.allclose(o0["after_zero_grad"], single.flat, rtol=1e-5, atol=1e-7)
    # two half-shard means summed = 2 x the shard mean
    assert torch.allclose(o0["accumulated"], 2 * single.flat, rtol=1e-5, atol=1e-7)
    assert torch.equal(o0["replayed"], torch.full_like(single.flat, 1.5))  # mean of rank + 1 over two ranks


# The engine's own step loop (SamplerTrainStep: begin_step / zero_grad / loss composition / reduce) under two ranks, with the
# CPU restatement of the module and an external task loss.
def _engine_worker(rank, world, port, x_all, state, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer, shard_batch

    torch.set_num_threads(1)
    net = _engine_net(state)
    red = FlatGradAllReducer(net)
    x = shard_batch(x_all, rank, world)
    step = SamplerTrainStep(net, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, task_loss=lambda p: (p ** 2).mean(),
                            reducer=red, use_graph=False, fused_loss=False)
    assert not step.in_graph and not step.split
    for _ in range(2):
        loss = step(x)
    torch.save((red.flat.clone(), loss.clone()), os.path.join(out_dir, "eng%d.pt" % rank))
    dist.destroy_process_group()


def _engine_net(state):
    from oracle.cpu_reference_model import SampleNetCPU

    class Net(SampleNetCPU):  # the attributes engine.SamplerTrainStep reads off a SampleNet
        skip_projection, input_shape, output_shape = False, "bnc", "bnc"

        def get_projection_loss(self):
            return self.sigma()

    net = Net(16, 32, 4)
    net.load_state_dict(state)
    net.eval()  # BatchNorm on running statistics: per-shard == whole-batch arithmetic
    net.training = True  # ... but the engine's step is the training step
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.training = False
    return net


def test_two_rank_engine_step(tmp_path):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.cpu_reference_model import SampleNetCPU
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    torch.manual_seed(1)
    ref = SampleNetCPU(16, 32, 4)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    x_all = torch.rand(8, 128, 3) - 0.5
    mp.spawn(_engine_worker, args=(2, _free_port(), x_all, state, str(tmp_path)), nprocs=2, join=True)
    (f0, l0), (f1, l1) = torch.load(tmp_path / "eng0.pt"), torch.load(tmp_path / "eng1.pt")
    assert torch.equal(f0, f1)
    net = _engine_net(state)
    red = FlatGradAllReducer(net)
    step = SamplerTrainStep(net, x_all, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, task_loss=lambda p: (p ** 2).mean(),
                            reducer=red, use_graph=False, fused_loss=False)
    loss = step(x_all)
    assert torch.allclose(f0, red.flat, rtol=1e-4, atol=1e-7)
    assert abs(float(l0 + l1) / 2 - float(loss)) < 1e-5


def _syncbn_comm_worker(rank, world, port, q):
    import os

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from samplenet_amd.syncbn import DistComm, combine_stats

        comm = DistComm()
        torch.manual_seed(7)
        rows = torch.randn(world * 5, 3, dtype=torch.float64) * torch.tensor([1.0, 1e-3, 10.0]) + torch.tensor([0.0, 100.0, -3.0])
        mine = rows[rank * 5:(rank + 1) * 5]
        packed = torch.stack([mine.mean(0), mine.var(0, unbiased=False), torch.full((3,), 5.0, dtype=torch.float64)])
        mean, var, total = combine_stats(comm.all_gather(packed), 1e-5)
        ok = (torch.allclose(mean, rows.mean(0), rtol=1e-12, atol=1e-12) and torch.allclose(var, rows.var(0, unbiased=False), rtol=1e-9, atol=1e-15)
              and float(total) == world * 5)
        s = comm.all_reduce_sum(torch.full((2, 3), float(rank + 1)))
        ok = ok and bool((s == sum(range(1, world + 1))).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_syncbn_communicator_and_statistics_merge_over_gloo():
    """syncbn.DistComm (all_gather / all_reduce SUM of the per-layer statistics) and the parallel-variance merge over two gloo
    ranks: the merged mean / biased variance equal those of the union of the ranks' rows, including a channel whose variance is
    1e-10 of its squared mean."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_syncbn_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _surface_listener_worker(rank, world, port, q):
    """VERDICT r4 #5a / ADVICE r4: under a multi-rank process group the captured module surface must step aside unless a
    FlatGradAllReducer is attached (DistributedDataParallel's reducer hooks hang on AccumulateGrad nodes the surface bypasses)."""
    import warnings

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from samplenet_amd import SampleNet, surface
        from samplenet_amd.parallel import FlatGradAllReducer

        net = SampleNet(8, 16, group_size=4, input_shape="bnc", output_shape="bnc")
        res = {}
        res["bare"] = surface.autograd_listeners(net)                      # 2 ranks, nothing attached: DDP must be assumed
        ddp = torch.nn.parallel.DistributedDataParallel(net)               # (CPU module: gloo DDP)
        res["ddp"] = surface.autograd_listeners(ddp.module)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            x = torch.rand(2, 32, 3)
            ok1 = surface._supported(net, x.requires_grad_(False))         # (CPU tensor: unsupported anyway, no warning needed)
            a = surface._fallback(net, "x")
            b = surface._fallback(net, "x")
            res["one_warning"] = (a, b, len([m for m in w if "op by op" in str(m.message)]))
        net.graph_surface = "force"
        res["force"] = surface.autograd_listeners(net)
        net.graph_surface = True
        red = FlatGradAllReducer(net, kernel_written=[n for n, _ in net.named_parameters() if not n.startswith("project")])
        res["reducer"] = surface.autograd_listeners(net)
        res["sink_backref"] = net._grad_sink.reducer is red
        h = net.fc1.weight.register_hook(lambda g: g)
        res["hook"] = surface.autograd_listeners(net)
        h.remove()
        res["hook_removed"] = surface.autograd_listeners(net)
        # reduce() right after a captured backward whose graph carried the collective: bookkeeping only, once
        red._graph_reduced = True
        before = red.flat.clone().fill_(float(rank + 1))
        red.flat.copy_(before)
        red.reduce()
        res["skipped_once"] = bool(torch.equal(red.flat, before)) and red._graph_reduced is False
        red.reduce()
        res["then_reduced"] = float(red.flat[0])
        # ... but a local (unreduced) contribution of the same step forces the collective
        red.flat.fill_(float(rank + 1))
        red._graph_reduced, red._unreduced = True, True
        red.reduce()
        res["mixed_reduced"] = float(red.flat[0])
        res["flags_cleared"] = (red._graph_reduced, red._unreduced) == (False, False)
        q.put((rank, res, ok1))
    finally:
        dist.destroy_process_group()


def test_captured_surface_steps_aside_under_ddp_and_hooks():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_surface_listener_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, ok1 in got:
        assert res["bare"] and "DistributedDataParallel" in res["bare"]
        assert res["ddp"] and "DistributedDataParallel" in res["ddp"]
        assert res["one_warning"] == (False, False, 1)
        assert res["force"] is None and res["reducer"] is None and res["sink_backref"]
        assert res["hook"] == "a parameter carries an autograd hook" and res["hook_removed"] is None
        assert res["skipped_once"] and abs(res["then_reduced"] - 1.5) < 1e-6  # mean of (1, 2) over the two ranks
        assert abs(res["mixed_reduced"] - 1.5) < 1e-6 and res["flags_cleared"]
        assert ok1 is False
