"""The RCCL path on ONE GPU (SURVEY 8e "test without a cluster"): torch.distributed backend "nccl" (= RCCL on ROCm) with
world_size 1 and FlatGradAllReducer(force_collective=True), so that all_reduce(AVG) really is issued on the flat gradient
bucket -- eager with the early (FC-head) collective on the side stream, and under the captured step, both as one graph +
one collective and as the split pair of graphs with the first collective between them.  With one rank AVG is the identity:
every variant must leave exactly the gradients of the collective-free step."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _nets(n):
    from samplenet_amd import SampleNet

    torch.manual_seed(0)
    net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    return [net] + [copy.deepcopy(net) for _ in range(n - 1)]


def test_eager_step_with_collectives(rccl):
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    na, nb, nc = _nets(3)
    x = torch.rand(32, 1024, 3, device="cuda") - 0.5
    ra = FlatGradAllReducer(na)                                         # no collective at world size 1
    rb = FlatGradAllReducer(nb, force_collective=True, overlap=True)    # FC-head bucket early on the side stream + the rest
    rc = FlatGradAllReducer(nc, force_collective=True, overlap=False)   # one collective over the whole bucket
    assert not ra.collective and rb.collective and rb.overlap and not rc.overlap
    for red, net in ((ra, na), (rb, nb), (rc, nc)):
        step = SamplerTrainStep(net, x, reducer=red, use_graph=False)
        for _ in range(2):
            loss = step(x)
        torch.cuda.synchronize()
        red.loss = float(loss)
    assert ra.loss == rb.loss == rc.loss
    assert torch.equal(ra.flat, rb.flat) and torch.equal(ra.flat, rc.flat)
    assert float(ra.flat.abs().sum()) > 0


@pytest.mark.parametrize("mode", ["graph", "graph-fork", "after", "split"])
def test_captured_step_with_collectives(rccl, mode):
    """Where the gradient collective of a captured step runs: INSIDE the step's one graph at its end (engine default), inside
    it with the FC-head segment forked to a side stream, from Python behind each replay, or between two graphs.  With one rank
    AVG is the identity: every placement must leave exactly the gradients of the collective-free step, on every ring entry,
    in any replay order."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    na, nb = _nets(2)
    ring_a = [torch.rand(32, 1024, 3, device="cuda") - 0.5 for _ in range(3)]
    ring_b = [t.clone() for t in ring_a]
    ra = FlatGradAllReducer(na)
    rb = FlatGradAllReducer(nb, force_collective=True)
    sa = SamplerTrainStep(na, ring_a[0], reducer=ra, input_ring=ring_a)
    overlap = mode == "split"
    sb = SamplerTrainStep(nb, ring_b[0], reducer=rb, input_ring=ring_b, overlap_allreduce=overlap,
                          allreduce="after" if overlap else mode)
    assert not sa.split and not sa.in_graph and sb.split == overlap
    assert sb.in_graph == (mode in ("graph", "graph-fork")) and sb.allreduce == ("after" if overlap else mode)
    assert all(len(g) == (2 if overlap else 1) for g in sb._ring_graphs)
    for i in (0, 1, 2, 1, 0, 0):
        la, lb = sa.replay(i), sb.replay(i)
        torch.cuda.synchronize()
        assert float(la) == float(lb), i
        assert torch.equal(ra.flat, rb.flat), i
    for (n, a), (_, b) in zip(na.named_buffers(), nb.named_buffers()):
        assert torch.equal(a, b), n  # BatchNorm running statistics moved identically


def test_allreduce_auto_picks_one_placement_and_stays_exact(rccl):
    """SamplerTrainStep(allreduce='auto') (VERDICT r5 #7b): the three placements of the gradient collective are each captured and
    timed at start-up, one is kept -- and whichever it is, the replays leave exactly the gradients of the collective-free step
    (one rank: AVG is the identity).  The probe's record names the choice and a finite time for every placement that could be
    captured."""
    import math

    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    na, nb = _nets(2)
    ring_a = [torch.rand(32, 1024, 3, device="cuda") - 0.5 for _ in range(2)]
    ring_b = [t.clone() for t in ring_a]
    ra = FlatGradAllReducer(na)
    rb = FlatGradAllReducer(nb, force_collective=True)
    sb = SamplerTrainStep(nb, ring_b[0], reducer=rb, input_ring=ring_b, allreduce="auto")
    probe = sb.allreduce_probe
    assert probe is not None and probe["chosen"] == sb.allreduce and sb.allreduce in ("graph", "after", "graph-fork")
    assert math.isfinite(probe["ms_per_step"][sb.allreduce]) and probe["ms_per_step"][sb.allreduce] < 1e8
    assert sb.in_graph == (sb.allreduce != "after") and len(sb._ring_graphs) == 2
    print("allreduce='auto' at world size 1 (collective forced):", probe)
    # (the probes ran extra steps on replica b: its running statistics are ahead; training-mode gradients do not read them)
    sa = SamplerTrainStep(na, ring_a[0], reducer=ra, input_ring=ring_a)
    for i in (0, 1, 1, 0):
        la, lb = sa.replay(i), sb.replay(i)
        torch.cuda.synchronize()
        assert float(la) == float(lb), i
        assert torch.equal(ra.flat, rb.flat), i
    # without a collective 'auto' is the plain captured step
    nc = _nets(1)[0]
    sc = SamplerTrainStep(nc, ring_a[0], reducer=FlatGradAllReducer(nc), input_ring=ring_a, allreduce="auto")
    assert sc.allreduce_probe is None and not sc.in_graph


def test_in_graph_collective_with_an_outside_task_loss(rccl):
    """The captured fused step with a task loss outside the node (proj differentiable) and the all-reduce inside the graph."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    na, nb = _nets(2)
    x = torch.rand(32, 1024, 3, device="cuda") - 0.5
    w = torch.randn(32, 64, 3, device="cuda")
    ra, rb = FlatGradAllReducer(na), FlatGradAllReducer(nb, force_collective=True)
    sa = SamplerTrainStep(na, x, reducer=ra, task_loss=lambda p: (p * w).mean())
    sb = SamplerTrainStep(nb, x, reducer=rb, task_loss=lambda p: (p * w).mean())
    assert sa._fast_path() and sb._fast_path() and sb.in_graph and not sa.in_graph
    for _ in range(3):
        la, lb = sa(x), sb(x)
    torch.cuda.synchronize()
    assert float(la) == float(lb) and torch.equal(ra.flat, rb.flat) and float(ra.flat.abs().sum()) > 0


def test_sync_batchnorm_over_rccl_at_world_size_one(rccl):
    """syncbn.convert_sync_batchnorm through the real communicator (all_gather / all_reduce over RCCL) at world size 1, where the
    statistics of "all ranks" are the local ones: the engine runs such a step eagerly (collectives between the layers), and loss and
    gradients agree with the per-process-statistics step within the rounding of the layer-by-layer kernels."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer
    from samplenet_amd.syncbn import convert_sync_batchnorm

    na, nb = _nets(2)
    convert_sync_batchnorm(nb)
    x = torch.rand(32, 1024, 3, device="cuda") - 0.5
    ra, rb = FlatGradAllReducer(na), FlatGradAllReducer(nb, force_collective=True)
    sa = SamplerTrainStep(na, x, reducer=ra)
    sb = SamplerTrainStep(nb, x, reducer=rb)
    assert sa._fast_path() and not sb._fast_path() and sb.graph is None
    la, lb = sa(x), sb(x)
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(la)))
    assert float((ra.flat - rb.flat).norm()) <= 2e-4 * float(ra.flat.norm())


def test_module_surface_with_the_collective_inside_its_backward_graph(rccl):
    """VERDICT r4 #5b: the reference call pattern under data parallelism -- net(x), the getters, backward(), reducer.reduce() -- on the
    captured module surface with the RCCL all-reduce as the LAST NODE of the backward graph (world size 1, collective forced: AVG is
    the identity, so every step must leave exactly the gradients of the collective-free surface)."""
    from samplenet_amd import surface
    from samplenet_amd.parallel import FlatGradAllReducer

    na, nb = _nets(2)
    ra, rb = FlatGradAllReducer(na), FlatGradAllReducer(nb, force_collective=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    xs = [torch.rand(32, 1024, 3, device="cuda", generator=g) - 0.5 for _ in range(6)]
    for i, x in enumerate(xs):
        for net, red in ((na, ra), (nb, rb)):
            red.zero_grad()
            simp, proj = net(x)
            loss = 0.01 * net.get_simplification_loss(x, simp, 64, 1.0, 0.0) + 0.01 * net.get_projection_loss() + proj.mean()
            loss.backward()
            red.reduce()
            net.last = float(loss)
        torch.cuda.synchronize()
        assert na.last == nb.last, i
        assert torch.equal(ra.flat, rb.flat), i
    pa, pb = surface.plans(na), surface.plans(nb)
    assert pa and pb and pb[0].reducer is rb and pb[0].collective_in_graph and not pa[0].collective_in_graph
    assert not rb._graph_reduced  # (consumed by reduce())
    for (n, a), (_, b) in zip(na.named_buffers(), nb.named_buffers()):
        assert torch.equal(a, b), n
