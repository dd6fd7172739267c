"""Worker of tests/test_gpu_two_ranks.py: ONE of two processes that share cuda:0, each with its own HIP SampleNet replica.

    python tests/_two_rank_worker.py <rank> <world> <port> <mode> <dir>

torch.distributed backend "gloo" (RCCL refuses two ranks on one device); the gradient bucket is a CUDA tensor -- where this
build's gloo cannot take device tensors the collective is staged through the host for the test (the reducer's own call
otherwise).  Modes: "fused" (captured fused step, per-rank statistics, allreduce='after'), "fixed" (running-statistics
BatchNorm: two ranks on half batches = one process on the whole batch), "sync" (convert_sync_batchnorm), "surface" (the
MODULE SURFACE as an unmodified script drives it -- net(x), the getters, backward(), reducer.reduce() -- on captured graphs
with the reducer's bucket), "ddp" (torch DistributedDataParallel around the module: the surface must step aside)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def gloo_takes_device_tensors():
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        return bool((t == dist.get_world_size()).all())
    except Exception:  # noqa: BLE001
        return False


def stage_collectives_through_the_host():
    ar, ag = dist.all_reduce, dist.all_gather

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return ar(t, op=op, group=group, async_op=async_op)
        h = t.detach().cpu()
        ar(h, op=op, group=group)
        t.copy_(h)

    def all_gather(out, t, group=None, async_op=False):
        if not t.is_cuda:
            return ag(out, t, group=group, async_op=async_op)
        hs = [o.detach().cpu() for o in out]
        ag(hs, t.detach().cpu(), group=group)
        for o, h in zip(out, hs):
            o.copy_(h)

    dist.all_reduce, dist.all_gather = all_reduce, all_gather


def fixed_stats_features(net):
    """The head with running-statistics BatchNorm while the module stays in training mode (projection + losses active)."""
    from samplenet_amd import pointnet

    def features(x, x_bnc=None):
        if x_bnc is None:
            x_bnc = x.permute(0, 2, 1)
        y = pointnet.PointNetMLPFunction.apply(net, x_bnc.contiguous(), False, *pointnet.param_list(net))
        return y.view(-1, 3, net.num_out_points)

    net._features = features
    net.graph_surface = False


def script_modes(mode, net, x, rank, world, out, native):
    """registration/main.py:507-531 as a script issues it, data parallel: (surface) FlatGradAllReducer + the captured module
    surface; (ddp) torch DistributedDataParallel -- the surface must step aside (one warning) and DDP's hooks synchronise."""
    import warnings

    from samplenet_amd import surface
    from samplenet_amd.parallel import FlatGradAllReducer

    M = net.num_out_points
    red, model = None, net
    if mode == "surface":
        red = FlatGradAllReducer(net)
    else:
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])
    captured = []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for it in range(5):
            if red is not None:
                red.zero_grad()
            else:
                for p in net.parameters():
                    p.grad = None
            simp, proj = model(x)
            loss = 0.01 * net.get_simplification_loss(x, simp, M, 1.0, 0.0) + 0.01 * net.get_projection_loss() + proj.mean()
            loss.backward()
            if red is not None:
                red.reduce()
            captured.append(bool(surface.plans(net)))
        nwarn = len([m for m in w if "op by op" in str(m.message)])
    torch.cuda.synchronize()
    flat = red.flat.cpu() if red is not None else torch.cat([p.grad.reshape(-1) for _, p in net.named_parameters()]).cpu()
    torch.save({"flat": flat, "loss": float(loss), "native_gloo": native, "captured": captured, "warnings": nwarn,
                "buffers": {k: v.cpu() for k, v in net.named_buffers()}}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    native = gloo_takes_device_tensors()
    if not native:
        stage_collectives_through_the_host()
    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer, shard_batch

    blob = torch.load(os.path.join(out, "input.pt"))
    net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    net.load_state_dict(blob["state"])
    x = shard_batch(blob["x"].cuda(), rank, world).contiguous()
    if mode == "sync":
        from samplenet_amd.syncbn import convert_sync_batchnorm

        convert_sync_batchnorm(net)
    elif mode == "fixed":
        fixed_stats_features(net)
    if mode in ("surface", "ddp"):
        script_modes(mode, net, x, rank, world, out, native)
        return
    red = FlatGradAllReducer(net)
    assert red.world == world and red.collective
    step = SamplerTrainStep(net, x, reducer=red, use_graph=(mode == "fused"), allreduce="after",
                            fused_loss=(mode != "fixed"))
    assert step._fast_path() == (mode == "fused")
    dist.barrier()
    for _ in range(3 if mode == "fused" else 1):  # both processes' chain kernels resident on the one device at once
        loss = step(x)
    torch.cuda.synchronize()
    step.check()  # FC chain hand-off error words: clean with two processes on the device
    torch.save({"flat": red.flat.cpu(), "loss": float(loss), "native_gloo": native,
                "buffers": {k: v.cpu() for k, v in net.named_buffers()}}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
