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
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz (dense fp32 matrix peak)
# The conv-stack GEMMs compute fp32 products as six bf16 products of three-way split operands on the bf16 matrix cores
# (pointnet_mlp.hip, gemm_tile_bx3 / conv_bwd_bx3_kernel): their matrix-pipe ceiling is the dense bf16 peak / 6
# (MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16; tools/micro/bf16x3_gemm.hip measures 392 fp32-equivalent TFLOP/s).
MFMA_SPLIT_BF16_PEAK_TFLOPS = 2500.0 / 6.0


def geometry_bytes_fwd(N, M, K):
    # fwd = 12N + 12M + 4MK + 12M + 8M + 8N  (P, Q in; knn idx, proj, dist1/idx1, dist2/idx2 out)
    return 12 * N + 12 * M + 4 * M * K + 12 * M + 8 * M + 8 * N


def time_pairscan_kernel(net, pool, K, reps=50):
    """Average duration of the geometric kernel of the step -- sn::pairscan_kernel as sn_pairscan_forward_partial launches
    it: kNN + soft projection + both Chamfer directions (the per-point side as partial keys) -- measured with HIP events on
    the stream it is launched on, on the bench's own inputs, launched back to back through the C ABI (ctypes: the host
    side stays ahead of the 11 us kernel)."""
    from samplenet_amd._lib import check, lib, ptr

    with torch.no_grad():
        x = pool[0]
        B, N, _ = x.shape
        y = net._features(x.permute(0, 2, 1), x).contiguous()  # (B,3,M)
        M = y.shape[2]
        dev = x.device
        G = lib.sn_pairscan_colmin_splits(B, N, M)
        T = net.project._temperature.detach().float().reshape(1)
        if G <= 1:  # batch so large that a cloud is one workgroup: the step then uses the finalising entry point
            from samplenet_amd import ops

            P, Q = x.contiguous(), y
            for _ in range(5):
                ops.SoftProjectFunction.apply(P, Q, T, 1e-2, K, True, ops.BNC, ops.BNC)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.SoftProjectFunction.apply(P, Q, T, 1e-2, K, True, ops.BNC, ops.BNC)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        proj = torch.empty(B, M, 3, device=dev)
        idx = torch.empty(B, M, K, device=dev, dtype=torch.int32)
        dq = torch.empty(B, M, device=dev)
        iq = torch.empty(B, M, device=dev, dtype=torch.int32)
        ws = torch.empty(B * max(G, 1) * N, device=dev, dtype=torch.int64)
        st = torch.cuda.current_stream().cuda_stream

        def launch():
            check(lib.sn_pairscan_forward_partial(B, N, M, K, ptr(x), 0, ptr(y), 1, ptr(idx), ptr(dq), ptr(iq), ptr(proj), 0,
                                                  ptr(T), 1e-2, ptr(ws), ws.numel() * 8, st), "sn_pairscan_forward_partial")

        for _ in range(5):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def time_conv5_backward_kernel(B, N, reps=50):
    """Average duration of the heaviest GEMM kernel of the step -- sn::conv_bwd_bx3_kernel<128,128,DZ_POOL>, the backward of
    the last 1x1 convolution (128 -> bottleneck 128 channels over B*N rows): data gradient + weight gradient from one
    pass -- measured with HIP events on the stream it is launched on, launches back to back, on tensors of the bench's
    shapes (values do not matter for its duration).  Algorithmic work per launch: 2*R*Ci*Co (dgrad) + 2*R*Ci*Co (wgrad)
    flop; algorithmic bytes: Z (R*Co*4) and Zprev (R*Ci*4) in, dYprev (R*Ci*4) out."""
    from samplenet_amd._lib import check, lib, ptr

    dev = torch.device("cuda", torch.cuda.current_device())
    R, Ci, Co = B * N, 128, 128
    g = torch.Generator(device=dev).manual_seed(7)
    z = torch.randn(R, Co, device=dev, generator=g)
    zprev = torch.randn(R, Ci, device=dev, generator=g)
    kc = torch.randn(3, Co, device=dev, generator=g)
    gsel = torch.randn(B, Co, device=dev, generator=g)
    argsel = torch.randint(0, N, (B, Co), device=dev, generator=g, dtype=torch.int32)
    W = torch.randn(Co, Ci, device=dev, generator=g) * 0.1
    coefp = torch.rand(4, Ci, device=dev, generator=g) + 0.5
    dyprev = torch.empty(R, Ci, device=dev)
    stats = torch.empty(lib.sn_linear_stats_blocks(R), 2, Ci, device=dev)
    part = torch.empty(lib.sn_linear_wgrad_splits(R, Ci, Co, 0) * Co * Ci, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        check(lib.sn_conv_backward_partials(R, Ci, Co, 2, None, ptr(z), ptr(kc), ptr(gsel), ptr(argsel), N, ptr(W), ptr(zprev),
                                            ptr(coefp), ptr(dyprev), ptr(stats), ptr(part), st), "sn_conv_backward_partials")

    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, 4.0 * R * Ci * Co, 4.0 * R * (Co + 2 * Ci)


def time_pairscan_saturated(K, Bsat=4096, N=1024, M=64, reps=10):
    """The same geometric kernel at a saturating batch ("report the HBM fraction at B=32 and at a saturating
    batch"): sn_pairscan_forward_ws on Bsat clouds (every cloud one workgroup, per-point minima finalised in the kernel)."""
    from samplenet_amd._lib import check, lib, ptr

    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.rand(Bsat, N, 3, device=dev, generator=g) - 0.5
    y = torch.rand(Bsat, 3, M, device=dev, generator=g) - 0.5
    T = torch.ones(1, device=dev)
    proj = torch.empty(Bsat, M, 3, device=dev)
    idx = torch.empty(Bsat, M, K, device=dev, dtype=torch.int32)
    dq, iq = torch.empty(Bsat, M, device=dev), torch.empty(Bsat, M, device=dev, dtype=torch.int32)
    dp, ip = torch.empty(Bsat, N, device=dev), torch.empty(Bsat, N, device=dev, dtype=torch.int32)
    wsb = lib.sn_pairscan_workspace_bytes(Bsat, N, M)
    ws = torch.empty(max(wsb // 8, 1), device=dev, dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        check(lib.sn_pairscan_forward_ws(Bsat, N, M, K, ptr(x), 0, ptr(y), 1, ptr(idx), None, ptr(dq), ptr(iq), ptr(dp), ptr(ip),
                                         ptr(proj), 0, None, ptr(T), 1e-2, ptr(ws) if wsb else None, wsb, st), "sn_pairscan_forward_ws")

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, Bsat


def time_module_surface(dev, B, N, M, K, steps=60):
    """Secondary leg: what a user of the drop-in module surface gets (registration/main.py:507-531 + 557-577), driver-timed:
      eager  -- plain  simp, proj = net(x); alpha*get_simplification_loss + lmbda*get_projection_loss + task; backward()
                launched op by op from Python with the frozen PCRNet's Chamfer loss as the task term (host-bound);
      graph  -- the same general (any task_loss) path captured once by engine.SamplerTrainStep and replayed.
    Not the headline value: the headline path fuses the benchmark's stand-in task term mean(proj) into the step."""
    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer
    from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss

    torch.manual_seed(0)
    net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").to(dev).eval()
    for p in pcr.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand(B, N, 3, device=dev, generator=g) - 0.5        # source cloud p1 (sampled)
    template = torch.rand(B, N, 3, device=dev, generator=g) - 0.5  # template cloud p0 (complete, NUM_SAMPLED_CLOUDS == 1)

    def task(proj):  # main.py:557-577 with the sampled source in place of p1
        return pcrnet_chamfer_loss(pcr, template, proj)[0]

    def eager_step():
        for p in net.parameters():
            p.grad = None
        simp, proj = net(x)
        loss = 0.01 * net.get_simplification_loss(x, simp, M, 1, 0) + 0.01 * net.get_projection_loss() + task(proj)
        loss.backward()
        return loss

    out = {}
    for name in ("eager", "graph"):
        if name == "graph":
            # a fresh replica with a gradient bucket (what a data-parallel user holds): the captured backward writes the MLP
            # gradients in place, and no autograd state of the eager leg (created on another stream) is alive during capture
            gnet = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
            gnet.load_state_dict(net.state_dict())
            st = SamplerTrainStep(gnet, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, task_loss=task,
                                  reducer=FlatGradAllReducer(gnet), use_graph=True)
            run = lambda: st(x)  # noqa: E731
        else:
            run = eager_step
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(loss).item()
        del loss
        out[name] = {"value": B * steps / dt, "unit": "point-clouds/s", "ms_per_step": dt / steps * 1e3}
    out["task_loss"] = "frozen PCRNet (bottleneck 1024) on (template 1024 pts, projected 64 pts) + Chamfer(projected, rotated template)"
    return out


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/: FETCH_SIZE x 2 + WRITE_SIZE, KiB, as
    MI355X_MICROARCH.md prescribes for gfx950), or None when no profile of that kernel is on disk."""
    for rnd in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
        try:
            with open(path) as f:
                table = json.load(f)
        except OSError:
            continue
        for name, row in table.items():
            if kernel_prefix in name and "hbm_traffic_bytes_per_launch" in row:
                return row["hbm_traffic_bytes_per_launch"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU (BASELINE config: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step kernel by kernel instead of replaying a hipGraph")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--no-module-surface", action="store_true", help="skip the secondary module-surface leg")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="N > 1: capture the step as two graphs and launch the FC-head segment's all-reduce between them on a side "
                         "stream (default: one graph, one collective after it -- measured faster at world size 1)")
    ap.add_argument("--no-probes", action="store_true",
                    help="profiling runs: only the timed steps (no roofline kernel probes, no cpu_baseline, no module-surface leg)")
    ap.add_argument("--force-collective", action="store_true",
                    help="world size 1 under torchrun: still issue the gradient all-reduce (single-GPU exercise of the RCCL path)")
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
    if world > 1 or (args.force_collective and "MASTER_PORT" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import samplenet_amd
    from samplenet_amd import SampleNet, ops
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    B, N, M, K = args.batch, 1024, 64, 8
    torch.manual_seed(0)  # identical replicas on every rank (registration/main.py:18 seeds 0 as well)
    net = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-2,
                    input_shape="bnc", output_shape="bnc").to(dev).train()
    reducer = FlatGradAllReducer(net, force_collective=args.force_collective)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # per-rank data shard
    pool = [torch.rand(B, N, 3, device=dev, generator=g) - 0.5 for _ in range(8)]
    # sampler loss weights of registration/src/sputils.py:53-59: alpha=0.01, lmbda=0.01, gamma=1, delta=0
    # the 8 resident batches are the step's input ring (a data loader would write its H2D copies into them): one graph per
    # entry, no copy into a staging buffer on the timed path
    train_step = SamplerTrainStep(net, pool[0], alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=reducer,
                                  use_graph=not args.no_graph, input_ring=pool, overlap_allreduce=args.overlap_allreduce)

    def step(i):
        return train_step.replay(i % len(pool))

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

    if rank == 0 and args.no_probes:
        print(json.dumps({"value": world * B * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "note": "--no-probes run"}), flush=True)
    elif rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        kern_ms = time_pairscan_kernel(net, pool, K)
        alg = geometry_bytes_fwd(N, M, K) * B
        achieved = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        conv_ms, conv_flop, conv_bytes = time_conv5_backward_kernel(B, N)
        conv_tf = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        conv_gbs = conv_bytes / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0
        # which roof is nearer: the split-bf16 matrix ceiling or HBM
        conv_hbm_bound = conv_gbs / HBM_PEAK_GBS >= conv_tf / MFMA_SPLIT_BF16_PEAK_TFLOPS
        sat_ms, Bsat = time_pairscan_saturated(K)
        sat_gbs = geometry_bytes_fwd(N, M, K) * Bsat / (sat_ms * 1e-3) / 1e9
        # MLP work of the whole step: 3 x 67.93 MFLOP per cloud (forward + data gradient + weight gradient)
        step_flop = 3 * 2 * 33_964_032 * B
        step_tf = step_flop / (ms * 1e-3) / 1e12
        out = {
            "metric": "point-clouds/sec fwd+bwd, Bx1024->64 soft-proj+Chamfer",
            "value": value, "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SampleNet sampler train step (fwd + simplification/projection "
                                   "losses + bwd), B=%d per GPU, 1024->64 points, K=8, bottleneck 128; no optimizer step" % B,
                       "batch_per_gpu": B, "global_batch": B * world, "n_in": N, "n_out": M, "group_size": K,
                       "parallelism": "dp%d" % world,
                       "grad_allreduce": ("none" if not reducer.collective else
                                          "flat bucket over RCCL: FC-head segment between the step's two graphs on a side stream, "
                                          "conv segment after" if train_step.split else "flat bucket over RCCL: one collective after the step"),
                       "execution": "eager launches" if args.no_graph else "whole step replayed as one hipGraph",
                       "mlp": "hand-written MFMA kernels: conv stack = fp32 via split-bf16 products (fp32-accurate), FC head = fp32 MFMA"},
            # the heaviest GEMM kernel of the step (most flops and most bytes of any launch): backward of the last 1x1 convolution.
            # Both roofs are reported; "bound" names the nearer one.
            "roofline": {"kernel": "sn::conv_bwd_bx3_kernel<128,128,DZ_POOL> (conv5 backward: dgrad + wgrad; fp32 products as "
                                   "six bf16 MFMAs of three-way split operands)",
                         "bound": "hbm" if conv_hbm_bound else "mfma",
                         "achieved": conv_gbs if conv_hbm_bound else conv_tf,
                         "peak": HBM_PEAK_GBS if conv_hbm_bound else MFMA_SPLIT_BF16_PEAK_TFLOPS,
                         "unit": "GB/s" if conv_hbm_bound else "TFLOP/s",
                         "frac": conv_gbs / HBM_PEAK_GBS if conv_hbm_bound else conv_tf / MFMA_SPLIT_BF16_PEAK_TFLOPS,
                         "traffic": pmc_traffic("conv_bwd_bx3_kernel<128, 128"),
                         "algorithmic_bytes_per_launch": conv_bytes, "algorithmic_flop_per_launch": conv_flop,
                         "avg_launch_ms": conv_ms,
                         "hbm": {"achieved": conv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": conv_gbs / HBM_PEAK_GBS},
                         "mfma": {"achieved": conv_tf, "peak": MFMA_SPLIT_BF16_PEAK_TFLOPS, "unit": "fp32-equivalent TFLOP/s",
                                  "frac": conv_tf / MFMA_SPLIT_BF16_PEAK_TFLOPS,
                                  "vs_fp32_mfma_peak": conv_tf / MFMA_F32_PEAK_TFLOPS},
                         "note": "matrix ceiling = dense bf16 MFMA peak / 6 products (tools/micro/bf16x3_gemm.hip: 392 "
                                 "fp32-equivalent TFLOP/s measured, 155 for the fp32 MFMA)"},
            # the geometric kernel of the path (SURVEY 8d's per-cloud byte count applies to it)
            "roofline_geometry": {"kernel": "sn::pairscan_kernel<16,true,true> (kNN + soft projection + both Chamfer directions)",
                                  "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("pairscan_kernel<16"),
                                  "algorithmic_bytes_per_launch": alg, "avg_launch_ms": kern_ms,
                                  "saturating_batch": {"batch": Bsat, "achieved": sat_gbs, "frac": sat_gbs / HBM_PEAK_GBS,
                                                       "clouds_per_s": Bsat / (sat_ms * 1e-3), "avg_launch_ms": sat_ms}},
            # the whole step against the fp32 matrix peak: MLP flops per step / step time (the geometric and scalar kernels, the
            # launch gaps and the dependency chain are all in the denominator)
            "step_mfma": {"flop_per_step": step_flop, "achieved": step_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": step_tf / MFMA_F32_PEAK_TFLOPS,
                          "note": "against the fp32 MFMA peak (the conv GEMMs run as split-bf16 products, ceiling %.0f; the FC "
                                  "head on the fp32 MFMA)" % MFMA_SPLIT_BF16_PEAK_TFLOPS},
        }
        if world == 1 and not args.no_module_surface:
            out["module_surface"] = time_module_surface(dev, B, N, M, K)
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_reference_model import time_cpu_baseline

            out["cpu_baseline"] = time_cpu_baseline(B, N, M, K, budget_s=args.cpu_budget)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 was busy with the roofline timing loops: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
