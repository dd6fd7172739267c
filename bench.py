#!/usr/bin/env python3
"""bench.py -- point-clouds/s of the SampleNet sampler training step (fwd + losses + bwd) on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Unit of work: one step of the sampler as registration/main.py:500-531 issues it
    simp, proj = sampler(x)
    L = 0.01 * sampler.get_simplification_loss(x, simp, 64, 1, 0) + 0.01 * sampler.get_projection_loss() + mean(proj)
    L.backward()                       (+ gradient all-reduce over RCCL when N > 1)
on a synthetic batch of B x 1024 x 3 clouds already resident in HBM; no optimizer step, no data loading
(that is the metric's definition: "fwd+bwd").  Workload = BASELINE.json configs[1]: B = 32 per GPU,
1024 -> 64 points, K = 8 (weak scaling: every rank processes its own B = 32 shard of the global batch).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.29 TB/s measured by a float4 copy)


def geometry_bytes_fwd(N, M, K):
    # fwd = 12N + 12M + 4MK + 12M + 8M + 8N  (P, Q in; knn idx, proj, dist1/idx1, dist2/idx2 out)
    return 12 * N + 12 * M + 4 * M * K + 12 * M + 8 * M + 8 * N


def time_pairscan_kernel(net, pool, K, reps=50):
    """Average duration of the geometric kernel (sn_pairscan_forward: kNN + soft projection + both Chamfer
    directions) measured with HIP events on the stream it is launched on, on the bench's own inputs, launches
    back to back (so the bracket holds kernel time, not Python launch overhead)."""
    from samplenet_amd import ops

    with torch.no_grad():
        x = pool[0]
        simp, _ = net(x)
        P = x.permute(0, 2, 1).contiguous()
        Q = simp.permute(0, 2, 1).contiguous()
        T = net.project._temperature.detach()
        for _ in range(5):
            ops.SoftProjectFunction.apply(P, Q, T, 1e-2, K, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.SoftProjectFunction.apply(P, Q, T, 1e-2, K, True)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU (BASELINE config: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step kernel by kernel instead of replaying a hipGraph")
    ap.add_argument("--torch-mlp", action="store_true", help="A/B: feature extractor through torch.nn instead of the HIP MLP")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the cpu_baseline leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import samplenet_amd
    from samplenet_amd import SampleNet, ops
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    B, N, M, K = args.batch, 1024, 64, 8
    torch.manual_seed(0)  # identical replicas on every rank (registration/main.py:18 seeds 0 as well)
    net = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-2,
                    input_shape="bnc", output_shape="bnc", use_hip_mlp=not args.torch_mlp).to(dev).train()
    reducer = FlatGradAllReducer(net)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # per-rank data shard
    pool = [torch.rand(B, N, 3, device=dev, generator=g) - 0.5 for _ in range(8)]
    # sampler loss weights of registration/src/sputils.py:53-59: alpha=0.01, lmbda=0.01, gamma=1, delta=0
    train_step = SamplerTrainStep(net, pool[0], alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=reducer,
                                  use_graph=not args.no_graph)

    def step(i):
        return train_step(pool[i % len(pool)])

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        kern_ms = time_pairscan_kernel(net, pool, K)
        alg = geometry_bytes_fwd(N, M, K) * B
        achieved = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "point-clouds/sec fwd+bwd, Bx1024->64 soft-proj+Chamfer",
            "value": value, "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SampleNet sampler train step (fwd + simplification/projection "
                                   "losses + bwd), B=%d per GPU, 1024->64 points, K=8, bottleneck 128; no optimizer step" % B,
                       "batch_per_gpu": B, "global_batch": B * world, "n_in": N, "n_out": M, "group_size": K,
                       "parallelism": "dp%d" % world, "grad_allreduce": "1 flat bucket, RCCL" if world > 1 else "none",
                       "execution": "eager launches" if args.no_graph else "whole step replayed as one hipGraph",
                       "mlp": "torch.nn (A/B)" if args.torch_mlp else "hand-written fp32 MFMA kernels"},
            "roofline": {"kernel": "sn::pairscan_kernel<16,true,true> (kNN + soft projection + both Chamfer directions)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": kern_ms,
                         "note": "geometric kernel of the path; the MLP GEMM kernels are listed in profiles/"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_reference_model import time_cpu_baseline

            out["cpu_baseline"] = time_cpu_baseline(B, N, M, K, budget_s=args.cpu_budget)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
