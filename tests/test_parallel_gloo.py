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
