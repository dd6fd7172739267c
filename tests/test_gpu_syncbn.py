"""Synchronised BatchNorm (samplenet_amd/syncbn.py) on ONE GPU: two "ranks" as two threads of this process, each with
its own replica and half of the batch, exchanging statistics through an in-process communicator -- against ONE replica on the whole
batch with the ordinary per-process statistics."""
import copy
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


class ThreadComm:
    """all_gather / all_reduce_sum among the threads of one process (the contract of syncbn.DistComm)."""

    class Shared:
        def __init__(self, world):
            self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def all_gather(self, t):
        torch.cuda.current_stream().synchronize()
        self.s.slots[self.rank] = t.clone()
        self.s.barrier.wait()
        out = torch.stack([x.clone() for x in self.s.slots])
        self.s.barrier.wait()
        return out

    def all_reduce_sum(self, t):
        return self.all_gather(t).sum(0)


def _run_ranks(fns):
    errs = []

    def wrap(f):
        def g():
            try:
                with torch.cuda.device(0):
                    f()
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
                raise
        return g

    ts = [threading.Thread(target=wrap(f)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs


@pytest.mark.parametrize("B,N", [(16, 256), (8, 64)])
def test_two_ranks_with_synchronised_statistics_equal_one_process(B, N):
    """forward_sync / backward_sync of two replicas on the two halves of a batch (threads; the autograd engine would run both
    ranks' nodes on its one device thread, so the pair is driven directly) against forward_impl / backward_impl of one replica on
    the whole batch: head outputs, gradients (the ranks' average = what the step's all-reduce leaves) and running statistics."""
    from samplenet_amd import SampleNet, pointnet
    from samplenet_amd.syncbn import backward_sync, convert_sync_batchnorm, forward_sync

    torch.manual_seed(B + N)
    ref = SampleNet(32, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    nets = [copy.deepcopy(ref) for _ in range(2)]
    x = torch.rand(2 * B, N, 3, device="cuda") - 0.5
    gy = torch.randn(2 * B, 96, device="cuda")
    with torch.cuda.device(0):
        y_ref, saved = pointnet.forward_impl(ref, x, True)
        g_ref = pointnet.backward_impl(ref, saved, gy / (2 * B))
    torch.cuda.synchronize()
    shared = ThreadComm.Shared(2)
    outs, grads = [None, None], [None, None]

    def rank(r):
        def f():
            comm = ThreadComm(shared, r)
            convert_sync_batchnorm(nets[r], comm=comm)
            xr = x[r * B:(r + 1) * B].contiguous()
            y, sv = forward_sync(nets[r], xr, comm)
            grads[r] = backward_sync(nets[r], sv, gy[r * B:(r + 1) * B].contiguous() / B, comm)  # (the rank's own mean loss)
            torch.cuda.synchronize()
            outs[r] = y
        return f

    _run_ranks([rank(0), rank(1)])
    got = torch.cat(outs)
    assert float((got - y_ref).abs().max()) <= 2e-5 * max(1.0, float(y_ref.abs().max()))
    gmax = max(float(g.abs().max()) for g in g_ref.values())
    for n, g in g_ref.items():
        avg = 0.5 * (grads[0][n] + grads[1][n])  # the step's gradient all-reduce (AVG)
        assert float((avg - g).abs().max()) <= 3e-4 * float(g.abs().max()) + 3e-6 * gmax, n
    for (n, a), (_, b0), (_, b1) in zip(ref.named_buffers(), nets[0].named_buffers(), nets[1].named_buffers()):
        assert torch.equal(b0, b1), n  # both ranks hold the same running statistics ...
        assert torch.allclose(a.float(), b0.float(), rtol=2e-5, atol=1e-6), n  # ... those of the whole batch


def test_one_rank_sync_equals_plain_statistics():
    from samplenet_amd import SampleNet
    from samplenet_amd.syncbn import convert_sync_batchnorm

    torch.manual_seed(0)
    a = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    b = convert_sync_batchnorm(copy.deepcopy(a), comm=ThreadComm(ThreadComm.Shared(1), 0))
    x = torch.rand(8, 128, 3, device="cuda") - 0.5
    sa, _ = a(x)
    sb, _ = b(x)
    sa.sum().backward(), sb.sum().backward()
    assert float((sa.detach() - sb.detach()).abs().max()) <= 1e-5
    gmax = max(float(p.grad.abs().max()) for p in a.parameters() if p.grad is not None)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.grad is not None:  # (a bias in front of a BatchNorm has an analytically zero gradient: bounded on the global scale)
            assert float((p.grad - q.grad).abs().max()) <= 2e-4 * float(p.grad.abs().max()) + 3e-6 * gmax, n
