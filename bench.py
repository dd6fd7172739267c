#!/usr/bin/env python3
"""bench.py -- point-clouds/s of the SampleNet sampler training step (fwd + losses + bwd) on MI355X.

    python bench.py --gpus N --steps K --warmup W          N > 1: starts its own N ranks (one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (the same ranks under an external launcher)

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

MIN_TIMED_S = 0.25  # the timed region never ends sooner, whatever --steps says (blocks of K steps are added)


def self_launch(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks here (one process per GPU over RCCL), pass
    rank 0's JSON line through, return the launcher's exit code."""
    import socket
    import subprocess

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and "--launcher-selftest" not in argv:
        print("bench.py --gpus %d: this node exposes %d GPU(s)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def csrc_sha16():
    """Hash of the kernel sources + the C ABI headers: what a committed profile is a profile OF (the GPU box has no .git)."""
    import hashlib

    h = hashlib.sha256()
    for d, pat in (("samplenet_amd/csrc", (".hip", ".h", ".cpp")), ("include", (".h",))):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith(pat):
                with open(os.path.join(ROOT, d, name), "rb") as f:
                    h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")


def profile_dir():
    for rnd in ROUNDS:
        d = os.path.join(ROOT, "profiles", rnd)
        if os.path.exists(os.path.join(d, "bench_graph_kernel_stats.csv")):
            return rnd, d
    return None, None


def _profile_file(name):
    """The newest committed profiles/rNN/<name> (None if no round holds it)."""
    for rnd in ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            return path
    return None


def profile_provenance():
    """Which committed rocprofv3 summary the `traffic` / longest-kernel fields come from, the kernel-source hash it was taken at
    (profiles/rNN/PROFILE_SRC_SHA, written by tools/gpu_profile.sh) and whether the sources have changed since."""
    rnd, d = profile_dir()
    if rnd is None:
        return {"dir": None, "stale": None}
    sha = None
    try:
        sha = open(os.path.join(d, "PROFILE_SRC_SHA")).read().split()[0]
    except OSError:
        pass
    now = csrc_sha16()
    out = {"dir": "profiles/" + rnd, "src_sha16": sha, "current_src_sha16": now, "stale": (sha != now) if sha else None}
    if out["stale"] or sha is None:
        print("bench.py: WARNING: %s was taken at kernel sources %s, this run has %s -- its PMC traffic / kernel ranking may not "
              "describe these kernels" % (out["dir"], sha, now), file=sys.stderr)
    return out


def longest_kernel_of_profile():
    """(short name, avg us per launch, launches per step) of the kernel with the largest per-step time in the committed
    rocprofv3 --kernel-trace --stats summary of `bench.py --no-probes`, or None."""
    import csv

    rnd, d = profile_dir()
    if rnd is None:
        return None
    rows = [r for r in csv.DictReader(open(os.path.join(d, "bench_graph_kernel_stats.csv"))) if "sn::" in r["Name"]]
    ref = [r for r in rows if "pairscan_kernel" in r["Name"]]
    if not rows or not ref:
        return None
    steps = int(ref[0]["Calls"])
    best = max(rows, key=lambda r: float(r["TotalDurationNs"]) if int(r["Calls"]) >= steps // 2 else 0.0)
    name = best["Name"].replace("void ", "").replace("sn::", "")
    return name[:name.index("(")] if "(" in name else name, float(best["AverageNs"]) / 1e3, max(1, round(int(best["Calls"]) / steps))


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (spec)
HBM_ACHIEVABLE_GBS = 6290.0  # ... of which a float4 copy reaches 6.29 TB/s (same guide): reported beside the spec in every HBM block


def hbm_block(gbs, **more):
    """An HBM roofline block: fraction of the spec peak (`frac`, the contract's field) and of what a copy kernel reaches."""
    d = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_achievable": gbs / HBM_ACHIEVABLE_GBS}
    d.update(more)
    return d


def step_algorithmic_bytes(B, N=1024):
    # SURVEY.md 8d per cloud: activations written once forward + read once backward (2 x 448 ch x N x 4 B), the cloud (12 N), the
    # geometry's 50,688 B (soft projection + Chamfer, forward + backward) -- 3.73 MB at N = 1024
    return (2 * 448 * N * 4 + 12 * N + 50_688) * B
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz (dense fp32 matrix peak)
# The conv-stack GEMMs compute fp32 products as six bf16 products of three-way split operands on the bf16 matrix cores
# (mlp_device.h gemm_tile_bx3 / pointnet_mlp_backward.hip conv_bwd_bx3_kernel): their matrix-pipe ceiling is the dense bf16 peak / 6
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


def _graph_replay_ms(fn, warm=3, reps=30):
    """fn() -- one full training step built from autograd ops -- captured once as a hipGraph and replayed; ms per replay
    (HIP events on the replay stream)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    import gc

    gc.collect()  # (graphs of earlier legs that sit in reference cycles must not be destroyed during the capture below)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):  # (the warm-up's stream: its per-stream scratch is reused)
        out = fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def _wall_ms(run, steps):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def time_module_surface(dev, B, N, M, K, steps=60, headline_ms=None):
    """Secondary legs: what a user of the drop-in module surface gets with a REAL task loss (registration/main.py:507-531 +
    557-577: frozen PCRNet + Chamfer on the projected points), driver-timed:
      eager_mean_proj -- the headline's own step (task term mean(proj)) through the plain module surface: simp, proj = net(x);
                alpha*get_simplification_loss + lmbda*get_projection_loss + proj.mean(); backward() -- what an unmodified
                train script gets without the engine: since round 4 those calls replay two captured graphs (surface.py);
      eager  -- the same with the PCRNet task loss (the task network's own launches stay op by op);
      eager_two_clouds -- the script's DEFAULT step (--num-sampled-clouds 2, main.py:516-524): source and template both sampled under
                one loss, the PCRNet task on the two 64-point projections; each sampler pass replays its own pair of graphs;
      eager_mean_proj_reducer -- eager_mean_proj with a FlatGradAllReducer attached and reduce() after backward(): the data-parallel
                form of the script's loop; under `--force-collective` the RCCL all-reduce is the last node of the surface's backward graph;
      op_by_op_mean_proj / op_by_op / op_by_op_two_clouds -- the legs above with graph_surface = False (every call launched from Python);
      graph  -- engine.SamplerTrainStep(task_loss=...) captured once and replayed: the fused single-node step with the task
                loss OUTSIDE the node -- proj is a differentiable output, the task gradient re-enters the loss backward as an
                explicit tensor (same scan, fc4 inside the scan, deferred tail as the headline);
      graph_general -- the same step on the op-by-op general path (fused_loss=False): what round 2 ran for any task loss;
      task_only -- the task network's own forward + backward (PCRNet on (1024, 64)-point clouds + Chamfer), captured: the
                part of `graph` that is not the sampler;
      general_path_mean_proj -- the headline's stand-in task mean(proj) routed through the external-gradient path
                (task_loss=lambda p: p.mean()): isolates what the mean(proj) specialisation of the headline is worth."""
    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer
    from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss

    torch.manual_seed(0)
    net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").to(dev).eval()
    for p in pcr.parameters():
        p.requires_grad_(False)
    pcr.static_weights()  # frozen for the sampler's training (main.py:272-277): constants of the graphs captured around it
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

    def replica():
        # a fresh replica with a gradient bucket (what a data-parallel user holds): the captured backward writes the MLP
        # gradients in place, and no autograd state of the eager leg (created on another stream) is alive during capture
        gnet = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
        gnet.load_state_dict(net.state_dict())
        return gnet

    def eager_mean_proj_step():  # the headline's unit of work through the plain module surface
        for p in net.parameters():
            p.grad = None
        simp, proj = net(x)
        loss = 0.01 * net.get_simplification_loss(x, simp, M, 1, 0) + 0.01 * net.get_projection_loss() + proj.mean()
        loss.backward()
        return loss

    def two_clouds_step():  # registration/main.py:516-524, the script's DEFAULT (--num-sampled-clouds 2): source AND template sampled
        for p in net.parameters():
            p.grad = None
        s1, p1 = net(x)
        l1 = net.get_simplification_loss(x, s1, M, 1, 0)
        s0, p0 = net(template)
        l0 = net.get_simplification_loss(template, s0, M, 1, 0)
        loss = 0.01 * 0.5 * (l1 + l0) + 0.01 * net.get_projection_loss() + pcrnet_chamfer_loss(pcr, p0, p1)[0]
        loss.backward()
        return loss

    out = {}
    # the script's calls op by op (graph_surface off: what every round before the captured surface measured) ...
    net.graph_surface = pcr.graph_surface = False
    ms, loss = _wall_ms(eager_mean_proj_step, max(steps, 200))
    assert torch.isfinite(loss).item()
    out["op_by_op_mean_proj"] = {"value": B / ms * 1e3, "unit": "point-clouds/s", "ms_per_step": ms}
    ms, loss = _wall_ms(eager_step, steps)
    assert torch.isfinite(loss).item()
    out["op_by_op"] = {"value": B / ms * 1e3, "unit": "point-clouds/s", "ms_per_step": ms}
    ms, loss = _wall_ms(two_clouds_step, steps)
    out["op_by_op_two_clouds"] = {"value": 2 * B / ms * 1e3, "unit": "sampled point-clouds/s (two sampler passes per step)", "ms_per_step": ms}
    # ... and the SAME unmodified calls on the captured surface (samplenet_amd/surface.py: two hipGraphs behind net(x), the loss
    # getters and backward(); the default)
    net.graph_surface = pcr.graph_surface = True  # (graphed.py: the frozen task network's term replays two graphs of its own)
    ms, loss = _wall_ms(eager_mean_proj_step, max(steps, 400))
    assert torch.isfinite(loss).item()
    net.check()
    from samplenet_amd import surface

    captured = bool(surface.plans(net))
    out["eager_mean_proj"] = {"value": B / ms * 1e3, "unit": "point-clouds/s", "ms_per_step": ms, "captured_surface": captured}
    ms, loss = _wall_ms(eager_step, max(steps, 200))
    assert torch.isfinite(loss).item()
    out["eager"] = {"value": B / ms * 1e3, "unit": "point-clouds/s", "ms_per_step": ms, "captured_surface": captured}
    ms, loss = _wall_ms(two_clouds_step, max(steps, 200))
    assert torch.isfinite(loss).item()
    out["eager_two_clouds"] = {"value": 2 * B / ms * 1e3, "unit": "sampled point-clouds/s (two sampler passes per step)", "ms_per_step": ms,
                               "captured_plans": len(surface.plans(net))}
    # ... and under data parallelism as a script issues it (round 5): a FlatGradAllReducer attached, reduce() after backward(); the
    # plan's bucket is the reducer's, and with a process group (bench.py --force-collective: RCCL at world size 1) the all-reduce is
    # the last node of the backward graph
    import torch.distributed as dist

    dnet = replica()
    force = dist.is_available() and dist.is_initialized()
    red = FlatGradAllReducer(dnet, force_collective=force)

    def dp_script_step():
        red.zero_grad()
        simp, proj = dnet(x)
        loss = 0.01 * dnet.get_simplification_loss(x, simp, M, 1, 0) + 0.01 * dnet.get_projection_loss() + proj.mean()
        loss.backward()
        red.reduce()
        return loss

    ms, loss = _wall_ms(dp_script_step, max(steps, 400))
    assert torch.isfinite(loss).item()
    dplans = surface.plans(dnet)
    out["eager_mean_proj_reducer"] = {"value": B / ms * 1e3, "unit": "point-clouds/s", "ms_per_step": ms, "captured_surface": bool(dplans),
                                      "collective": bool(red.collective),
                                      "collective_in_backward_graph": bool(dplans and dplans[0].collective_in_graph)}
    del dnet, red
    for name, kw in (("graph", dict(task_loss=task)), ("graph_general", dict(task_loss=task, fused_loss=False)),
                     ("general_path_mean_proj", dict(task_loss=lambda p: p.mean()))):
        gnet = replica()
        st = SamplerTrainStep(gnet, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=FlatGradAllReducer(gnet),
                              use_graph=True, **kw)
        ms, loss = _wall_ms(lambda: st(x), max(steps, 200))
        assert torch.isfinite(loss).item(), name
        st.check()
        out[name] = {"value": B / ms * 1e3, "unit": "point-clouds/s", "ms_per_step": ms, "fast_path": bool(st._fast_path())}
        del st, gnet
    # the task network alone (forward + gradient to the 64 projected points), captured
    q = (torch.rand(B, M, 3, device=dev, generator=g) - 0.5).requires_grad_(True)

    def task_step():
        q.grad = None
        t = task(q)
        t.backward()
        return t

    tms, _ = _graph_replay_ms(task_step)
    out["task_only"] = {"ms_per_step": tms}
    if headline_ms:
        out["graph"]["vs_headline_plus_task"] = out["graph"]["ms_per_step"] / (headline_ms + tms)
    out["task_loss"] = "frozen PCRNet (bottleneck 1024) on (template 1024 pts, projected 64 pts) + Chamfer(projected, rotated template)"
    try:
        out["task_roofline"] = _task_wide_layer_roofline(dev, B, N)
    except Exception as e:  # noqa: BLE001
        out["task_roofline"] = {"error": repr(e)[:200]}
    return out


def _task_wide_layer_roofline(dev, B, N, Ci=128, Co=1024, reps=40):
    """The task step's longest kernel against its roof: PCRNet's 128 -> 1024 layer on the template cloud + max over the points
    (sn_linear_forward_maxpool_wide, task_network.hip) -- 2 R Ci Co fp32-equivalent flops as split-bf16 products (ceiling
    2.5 PF / 6 = 417 TFLOP/s), HIP events around back-to-back launches on the launch stream."""
    from samplenet_amd._lib import check, lib, ptr

    R = B * N
    if not lib.sn_linear_forward_maxpool_wide_supported(R, Ci, Co, N):
        return {"skipped": "shape not on the wide kernel"}
    gen = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(R, Ci, device=dev, generator=gen)
    coef = torch.zeros(4, Ci, device=dev)
    coef[0] = 1
    W = torch.randn(Co, Ci, device=dev, generator=gen) * 0.1
    b = torch.randn(Co, device=dev, generator=gen)
    pooled = torch.empty(B, Co, device=dev)
    planes = torch.empty(3 * Co * Ci, device=dev, dtype=torch.bfloat16)
    scratch = torch.empty(lib.sn_linear_forward_maxpool_wide_scratch_bytes(R, Ci, Co, N) // 8, device=dev, dtype=torch.int64)
    st = torch.cuda.current_stream(dev).cuda_stream

    def run(ready):
        check(lib.sn_linear_forward_maxpool_wide(R, Ci, Co, N, ptr(a), ptr(coef), ptr(W), ptr(b), None, ptr(scratch), ptr(pooled), None,
                                                 None, ptr(planes), ready, st), "sn_linear_forward_maxpool_wide")

    run(0)
    for _ in range(5):
        run(1)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(1)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * R * Ci * Co
    ach = flop / (ms * 1e-3) * 1e-12
    return {"kernel": "linear_fwd_wide_pool_kernel<128> + key decode (PCRNet conv5 on the template cloud, max over the points fused)",
            "bound": "mfma", "achieved": ach, "peak": 2500.0 / 6.0, "unit": "fp32-equivalent TFLOP/s (six bf16 products per fp32 product)",
            "frac": ach / (2500.0 / 6.0), "avg_launch_ms": ms, "algorithmic_flop_per_launch": flop,
            "algorithmic_bytes_per_launch": float(R * Ci * 4 + Co * Ci * 4 + B * Co * 4)}


def time_config3_emd(dev, reps=5):
    """BASELINE configs[3] (ShapeNet reconstruction, EMD loss): B = 50 clouds, approx_match / match_cost between the 2048-point
    reconstruction and its 2048-point target (reconstruction/src/samplenet_pointnet_ae.py:118-131; kernels
    classification/structural_losses/tf_approxmatch_g.cu).  Two forms, forward + gradients:
      emd_loss   -- ops.emd_loss = sn_emd_loss_fast: the auction + ONE sweep over 64 x 64 tiles that re-evaluates match from the per-level
                    ratio vectors once per pair for cost, grad1 and grad2, with the reference op's own __expf (v_exp_f32); the
                    (B, 2048, 2048) match matrix (839 MB) is never written
                    (`emd_loss_exact`: the same with approx_match's compensated exponential, bit-identical to three_call's cost);
      three_call -- approx_match -> match_cost -> its gradient, the reference's op sequence (match written once, read twice).
    Roofs: HBM on SURVEY 8d's algorithmic bytes (3 x 16.78 MB per cloud: the materialised form's minimum) and VALU issue --
    the kernels are exponential-bound: 3 exp per pair and level x 10 levels + 10 per pair (three_call's materialisation; emd_loss's sweep);
    256 CUs x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s, v_exp_f32 priced at 5/3 of a plain VALU op
    (MI355X_MICROARCH.md).  `valu_busy_profiled`: SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES of the committed rocprofv3 --pmc pass."""
    from samplenet_amd import ops

    B, n, m = 50, 2048, 2048
    g = torch.Generator(device=dev).manual_seed(3)
    a = (torch.rand(B, n, 3, device=dev, generator=g) - 0.5).requires_grad_(True)
    b = (torch.rand(B, m, 3, device=dev, generator=g) - 0.5).requires_grad_(True)

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_fused = timed(lambda: torch.autograd.grad(ops.emd_loss(a, b).sum(), [a, b]))
    t_exact = timed(lambda: torch.autograd.grad(ops.emd_loss(a, b, True).sum(), [a, b]))
    t_three = timed(lambda: torch.autograd.grad(ops.match_cost(a, b, ops.approx_match(a, b)).sum(), [a, b]))
    t_match = timed(lambda: ops.approx_match(a, b))
    pairs = float(B) * n * m
    alg = 3.0 * pairs * 4.0  # write match once + read it by cost and by grad
    valu_peak = 256 * 4 * 32 * 2.4e9
    out = {"workload": "BASELINE configs[3]: EMD loss fwd+grad, B=50, n=m=2048 (reconstruction/src/samplenet_pointnet_ae.py:118-131)"}
    sq = None
    try:
        rnd, d = profile_dir()
        sq = json.load(open(_profile_file("emd_sq_counters.json")))
    except (OSError, ValueError, TypeError):
        pass
    # (emd_loss since round 6: 30 exponentials per pair in the auction + 10 in ONE cost / gradient sweep -- each pair's match value is
    #  evaluated once for cost, grad1 and grad2 (emd_loss_sweep2d_kernel); until round 5 two sweeps: 50)
    for name, ms, exps, traffic_key in (("emd_loss", t_fused, 40.0, "emd_loss"), ("three_call", t_three, 40.0, "three_call")):
        lane_ops = pairs * (exps * (5.0 / 3.0) + 10 * 3 * 9.0 + 10.0)  # exps + ~9 packed-pair VALU slots per pair, level and pass
        gbs = alg / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "clouds_per_s": B / (ms * 1e-3),
                     "roofline": hbm_block(gbs, algorithmic_bytes=alg, traffic=pmc_traffic_sum(traffic_key),
                                           note="SURVEY 8d's minimum for the materialised form; emd_loss moves no match matrix at all"),
                     "valu_issue": {"exp_per_pair": exps, "model_lane_ops": lane_ops, "achieved_lane_ops_per_s": lane_ops / (ms * 1e-3),
                                    "peak_lane_ops_per_s": valu_peak, "frac": lane_ops / (ms * 1e-3) / valu_peak,
                                    "valu_busy_profiled": (sq or {}).get(name)}}
    out["emd_loss_exact"] = {"ms": t_exact, "clouds_per_s": B / (t_exact * 1e-3)}
    out["approx_match_only_ms"] = t_match
    return out


def pmc_traffic_sum(key):
    """HBM bytes per call of an EMD form from the committed PMC passes (profiles/rNN/emd_pmc_summary.json), or None."""
    try:
        return json.load(open(_profile_file("emd_pmc_summary.json"))).get(key)
    except (OSError, ValueError, TypeError):
        return None


def time_config3_sampler(dev, steps=40, K=16):
    """BASELINE configs[3], the sampler's side (the EMD leg above is the loss side): the reconstruction task's SampleNet
    (reconstruction/src/samplers.py:23-38: conv widths 64, 128, 128, 256, bottleneck 128, FC 256, 256 without BatchNorm) on
    B = 50 clouds of 2048 points -> 64, K = 16 (the reference's projection group size for this task: reconstruction/sampler/
    train_samplenet.py:50): forward + simplification / projection losses + backward, eager and captured.  Its
    128 -> 256 -> 128 layers run the fused backward kernel in two passes over the halves of the 256-channel side; the forward runs on
    the one-call statistics chain (sn_conv_stack_forward_bn: two accumulator blocks per layer, K = 256 on the pre-split planes)."""
    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    B, N, M = 50, 2048, 64
    torch.manual_seed(0)
    net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc", conv_widths=(64, 128, 128, 256), fc_widths=(256, 256),
                    fc_batchnorm=False, temperature_floor=1e-2, min_sigma=0.0).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(13)
    x = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    out = {"workload": "BASELINE configs[3], sampler side: reconstruction SampleNet (conv 64-128-128-256-128, FC 256-256 no BN), B=50, "
                       "2048->64, K=%d%s, fwd + losses + bwd" % (K, " (the reference's default: reconstruction/sampler/train_samplenet.py:50)" if K == 16 else "")}
    for name, use_graph in (("eager", False), ("graph", True)):
        import copy

        rep = copy.deepcopy(net)
        st = SamplerTrainStep(rep, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=FlatGradAllReducer(rep), use_graph=use_graph)
        ms, loss = _wall_ms(lambda: st(x), steps if not use_graph else max(steps, 100))
        assert torch.isfinite(loss).item(), name
        out[name] = {"ms_per_step": ms, "value": B / ms * 1e3, "unit": "point-clouds/s", "fast_path": bool(st._fast_path())}
        del st, rep
    # the same step as an unmodified script issues it (net(x), the two getters, backward()): on the captured module surface
    # since round 5 (surface.py: the sampler variants outside the registration architecture)
    import copy

    from samplenet_amd import surface

    rep = copy.deepcopy(net)

    def script():
        for p in rep.parameters():
            p.grad = None
        simp, proj = rep(x)
        loss = 0.01 * rep.get_simplification_loss(x, simp, M, 1.0, 0.0) + 0.01 * rep.get_projection_loss() + proj.mean()
        loss.backward()
        return loss

    ms, loss = _wall_ms(script, max(steps, 100))
    assert torch.isfinite(loss).item()
    out["script"] = {"ms_per_step": ms, "value": B / ms * 1e3, "unit": "point-clouds/s", "captured_surface": bool(surface.plans(rep))}
    return out


def time_config1_classification(dev, steps=100):
    """BASELINE configs[1]'s LITERAL network: the classification task's sampler (classification/models/samplenet_model.py:30-108:
    the registration widths plus a BatchNorm WITHOUT activation on the head's output, fc14b; projection group size 7:
    classification/train_samplenet.py:46; sigma = T^2 without a floor: classification/soft_projection.py:41), B = 32,
    1024 -> 64: forward + simplification / projection losses + backward.  `graph` = the whole step replayed as one hipGraph
    (engine.SamplerTrainStep, as the headline), `eager` = the same launches one by one, `script` = the reference call pattern
    (net(x), the two getters, backward()) through the plain module surface (captured graphs since round 6).  The headline above
    runs the registration sampler (the module the drop-in surface mirrors; K = 8); this leg is the same step with the output
    BatchNorm and K = 7: the head ends in sn_layer_forward_bn_out (fc4 + BatchNorm, one launch), the scan reads the queries, the
    backward opens with sn_bn_output_backward -- two launches more than the headline's step.  (With random-init weights the
    normalised queries spread like N(0, 1) around a cloud in [-0.5, 0.5]^3: a handful of queries own most of the points, and the
    loss backward -- one wave per query, its points summed in index order as the reference's Chamfer backward does -- runs
    ~24 us against ~11 us in the headline.)"""
    import copy

    from samplenet_amd import SampleNet, surface
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    B, N, M, K = 32, 1024, 64, 7
    torch.manual_seed(0)
    net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc", last_fc_batchnorm=True, min_sigma=0.0).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(21)
    x = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    out = {"workload": "BASELINE configs[1], literal network: classification SampleNet (BatchNorm on the head's output), B=32, "
                       "1024->64, K=7, fwd + losses + bwd"}
    for name, use_graph in (("eager", False), ("graph", True)):
        rep = copy.deepcopy(net)
        # (as the headline: the batch is resident in the step's input ring -- no copy into a staging buffer on the timed path)
        st = SamplerTrainStep(rep, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=FlatGradAllReducer(rep), use_graph=use_graph,
                              input_ring=[x])
        ms, loss = _wall_ms(lambda: st.replay(0), steps if not use_graph else max(steps, 300))
        assert torch.isfinite(loss).item(), name
        st.check()
        out[name] = {"ms_per_step": ms, "value": B / ms * 1e3, "unit": "point-clouds/s", "fast_path": bool(st._fast_path())}
        del st, rep
    rep = copy.deepcopy(net)

    def script():
        for p in rep.parameters():
            p.grad = None
        simp, proj = rep(x)
        loss = 0.01 * rep.get_simplification_loss(x, simp, M, 1.0, 0.0) + 0.01 * rep.get_projection_loss() + proj.mean()
        loss.backward()
        return loss

    ms, loss = _wall_ms(script, max(steps, 300))
    assert torch.isfinite(loss).item()
    out["script"] = {"ms_per_step": ms, "value": B / ms * 1e3, "unit": "point-clouds/s", "captured_surface": bool(surface.plans(rep))}
    return out


def time_config5_progressive(dev, steps=40):
    """BASELINE configs[4] on one GPU (per-rank work of the DP job): progressive SampleNet 1024 -> {32, 64, 128, 256} + the PCRNet
    registration task (classification/train_samplenet_progressive.py:157-234 for the prefix semantics, registration/main.py:
    507-531,557-577 for the task): the largest set sampled and projected once, the task network fed every prefix of the
    projected points, simplification loss summed over the prefixes; B = 32.  `eager` = as a script issues it: the sampler's calls on
    the captured module surface (the prefix losses' gradient on the simplified cloud is an operand of its backward graph), the frozen
    task network's four evaluations on graphed.py's graphs, the prefix losses as launches; `graph` = the whole step captured by hand."""
    from samplenet_amd import SampleNetProgressive
    from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss_multi

    B, N, K, sizes = 32, 1024, 8, [32, 64, 128, 256]
    torch.manual_seed(0)
    net = SampleNetProgressive(sizes, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").to(dev).eval()
    for p in pcr.parameters():
        p.requires_grad_(False)
    pcr.static_weights()  # frozen for the sampler's training (main.py:272-277): constants of the graphs captured around it
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    template = torch.rand(B, N, 3, device=dev, generator=g) - 0.5

    def step():
        for p in net.parameters():
            p.grad = None
        simp, proj = net(x)
        loss = 0.01 * net.get_progressive_simplification_loss(x, simp, 1, 0, "sum") + 0.01 * net.get_projection_loss()
        # the four evaluations share the template cloud: its extractor pass runs once, and the trunk once on all 4 x 32 rows
        # (the frozen network's whole call -- extractor, trunk, heads, rotations, Chamfer terms -- replays captured graphs: graphed.py)
        for task, _, _ in pcrnet_chamfer_loss_multi(pcr, template, net.prefixes(proj)):
            loss = loss + task
        loss.backward()
        return loss

    ems, loss = _wall_ms(step, steps)
    assert torch.isfinite(loss).item()
    # a fresh replica for the capture: no autograd state of the eager leg (AccumulateGrad nodes created on the default stream)
    # may be alive while a graph is being captured
    import copy

    del loss
    net = copy.deepcopy(net)
    gms, loss = _graph_replay_ms(step)
    assert torch.isfinite(loss).item()
    return {"workload": "BASELINE configs[4] per-rank: progressive SampleNet 1024 -> {32,64,128,256}, K=8, B=32, PCRNet + Chamfer "
                        "task on every prefix, fwd + losses + bwd",
            "template_features": "computed once per step, shared by the four task evaluations (PCRNet.template_features); their FC trunks run as one pass on 4 x 32 rows (pcrnet_chamfer_loss_multi)",
            "eager": {"ms_per_step": ems, "value": B / ems * 1e3, "unit": "point-clouds/s"},
            "graph": {"ms_per_step": gms, "value": B / gms * 1e3, "unit": "point-clouds/s"}}


def time_batch_sweep(dev, N, M, K, batches=(32, 48, 96, 128, 512, 2048)):
    """The whole sampler step (the headline's unit of work) at growing batches: B = 32 is latency-bound by construction (14
    dependent launches), larger batches show what the kernels reach when fed; 48 and 96 are batches that are not whole 64-row
    blocks (above 32 clouds the head runs layer by layer: row-blocked kernels).  Fractions: MLP flops / step time against the
    fp32 MFMA peak; SURVEY 8d's algorithmic bytes (3.73 MB per cloud: activations written once and read once per consumer,
    geometry 50.7 KB) / step time against 8 TB/s."""
    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    out = []
    for B in batches:
        torch.manual_seed(0)
        net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
        g = torch.Generator(device=dev).manual_seed(77)
        x = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
        st = SamplerTrainStep(net, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=FlatGradAllReducer(net), use_graph=True)
        ms, loss = _wall_ms(lambda: st(x), max(20, min(300, int(6400 / B))))
        assert torch.isfinite(loss).item()
        flop = 3 * 2 * 33_964_032 * B
        byts = step_algorithmic_bytes(B, N)
        out.append({"batch": B, "ms_per_step": ms, "clouds_per_s": B / ms * 1e3, "fused_single_node_step": bool(st._fast_path()),
                    "mfma_frac_fp32_peak": flop / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                    "hbm_frac_algorithmic": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "hbm_frac_algorithmic_of_achievable": byts / (ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS})
        del st, net
    return out


def time_fc_chain_backward(net, x, reps=20, inner=20):
    """Average duration of sn::fc_chain_bwd_kernel (the FC head's backward as one launch) on the bench's shapes: `inner` launches
    captured into a hipGraph (the host side allocates its outputs per call and would not keep ahead of a ~25 us kernel) and
    replayed, HIP events around the replays.  -> (ms per launch, algorithmic bytes, flop)."""
    from samplenet_amd import pointnet

    B = x.shape[0]
    with torch.no_grad():
        _, saved = pointnet.forward_impl(net, x, True)
        convs, fcs = pointnet._layers(net)
        gy = torch.randn(B, fcs[-1].Co, device=x.device)

        def launch():
            for _ in range(inner):
                if pointnet._fc_chain_bwd(net, convs, fcs, saved, gy, None, {}, False) is False:
                    raise RuntimeError("fc chain backward not supported at this shape")

        ms, _ = _graph_replay_ms(launch, warm=2, reps=reps)
    wbytes = sum(L.Co * L.Ci for L in fcs) * 4
    act = sum(B * (L.Co + L.Ci) for L in fcs) * 4
    flop = sum(4.0 * B * L.Co * L.Ci for L in fcs)  # data gradient + weight gradient
    return ms / inner, 2 * wbytes + act, flop


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/: FETCH_SIZE x 2 + WRITE_SIZE, KiB, as
    MI355X_MICROARCH.md prescribes for gfx950), or None when no profile of that kernel is on disk."""
    for rnd in ROUNDS:
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


def pmc_traffic_step():
    """Sum over the step's kernels of the committed PMC traffic per launch (one launch each per step at B = 32), or None."""
    for rnd in ROUNDS:
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")) as f:
                table = json.load(f)
        except OSError:
            continue
        tot = sum(row.get("hbm_traffic_bytes_per_launch", 0.0) for row in table.values())
        if tot > 0:
            return tot
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
    ap.add_argument("--no-module-surface", action="store_true", help="skip the secondary module-surface legs")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip config3_emd / config5_progressive / batch_sweep")
    ap.add_argument("--allreduce", choices=("auto", "graph", "graph-fork", "after", "split"), default="auto",
                    help="N > 1, where the gradient collective runs: inside the step's graph at its end, inside it with the FC-head "
                         "segment forked to a side stream, from Python after each replay, or between two graphs; auto (default): "
                         "the first three are each captured and timed for 20 replays at start-up, the fastest is kept")
    ap.add_argument("--overlap-allreduce", action="store_true", help="same as --allreduce split")
    ap.add_argument("--no-probes", action="store_true",
                    help="profiling runs: only the timed steps (no roofline kernel probes, no cpu_baseline, no secondary legs)")
    ap.add_argument("--force-collective", action="store_true",
                    help="world size 1: still issue the gradient all-reduce (single-GPU exercise of the RCCL path)")
    ap.add_argument("--min-time", type=float, default=MIN_TIMED_S, help="minimum length of the timed region in seconds")
    ap.add_argument("--only-leg", default=None, help="debugging: run only this secondary leg besides the headline")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no measurement: the N ranks only rendezvous (gloo, CPU), all-reduce their ranks and rank 0 prints a JSON "
                         "line -- exercises the --gpus N self-launch path on a host without GPUs (tests/)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and (args.gpus > 1 or args.force_collective):
        # `python bench.py --gpus N` as the driver contract spells it: start the ranks ourselves
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    if args.launcher_selftest:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rank_sum": float(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import samplenet_amd  # noqa: F401
    from samplenet_amd import SampleNet
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
    split = args.overlap_allreduce or args.allreduce == "split"
    train_step = SamplerTrainStep(net, pool[0], alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=reducer,
                                  use_graph=not args.no_graph, input_ring=pool, overlap_allreduce=split,
                                  allreduce="after" if split else args.allreduce)

    def step(i):
        return train_step.replay(i % len(pool))

    # warm-up (also the estimate that sizes the timed region: never shorter than --min-time, whatever --steps says)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    est = (time.perf_counter() - tw) / max(args.warmup, 1)
    blocks = 1
    if args.warmup > 0 and est * args.steps < args.min_time:
        blocks = int(args.min_time / (est * args.steps)) + 1
    if world > 1:  # every rank must run the same number of steps
        t = torch.tensor([blocks], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = int(t.item())
    total_steps = args.steps * blocks
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(total_steps):
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
    train_step.check()  # FC-chain hand-off error words (a timed-out launch would also have left a NaN loss)
    # the gradient collective ALONE (every rank: it is a collective), outside the timed region: what one all-reduce of the flat
    # bucket costs back to back at this world size -- the number the scaling curve is read against (a 0.18 ms step leaves no
    # room for a fully exposed 30-40 us collective at N = 8)
    allreduce_alone_us = None
    if reducer.collective:
        for _ in range(5):
            reducer._all_reduce_mean(reducer.flat)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            reducer._all_reduce_mean(reducer.flat)
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / 50 * 1e3], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_alone_us = float(t.item())

    if not train_step.split and train_step.in_graph:
        ar = "flat 1 MB bucket over RCCL, captured INSIDE the step's graph (%s)" % (
            "FC-head segment forked to a side stream, joined at the end" if train_step.allreduce == "graph-fork" else "one collective at its end")
    elif train_step.split:
        ar = "flat bucket over RCCL: FC-head segment between the step's two graphs on a side stream, conv segment after"
    else:
        ar = "flat bucket over RCCL: one collective launched after each step"
    if train_step.allreduce_probe is not None:
        ar += " -- chosen by allreduce='auto' (ms per step at start-up, slowest rank: %s)" % ", ".join(
            "%s %.4f" % kv for kv in train_step.allreduce_probe["ms_per_step"].items())
    if rank == 0 and args.no_probes:
        print(json.dumps({"value": world * B * total_steps / dt, "ms_per_step": dt / total_steps * 1e3, "n_gpus": world,
                          "grad_allreduce": ar if reducer.collective else "none",
                       "allreduce_alone_us": allreduce_alone_us, "allreduce_bytes": int(reducer.flat.numel() * 4), "allreduce_alone_us": allreduce_alone_us,
                          "note": "--no-probes run"}), flush=True)
    elif rank == 0:
        ms = dt / total_steps * 1e3
        value = world * B * total_steps / dt
        prov = profile_provenance()
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
        # the LONGEST kernel of the step per the committed rocprofv3 summary (not the heaviest): measured live when it is one
        # this file has a probe for
        longest = longest_kernel_of_profile()
        longest_out = None
        if longest is not None:
            lname, lavg_us, lcalls = longest
            longest_out = {"kernel": lname, "profile_avg_launch_us": lavg_us, "launches_per_step": lcalls, "profile": prov["dir"]}
            if lname.startswith("fc_chain_bwd_kernel") and B <= 32:
                fms, fbytes, fflop = time_fc_chain_backward(net, pool[0])
                fg, ft = fbytes / (fms * 1e-3) / 1e9, fflop / (fms * 1e-3) / 1e12
                longest_out.update(hbm_block(fg))
                longest_out.update({"avg_launch_ms": fms, "algorithmic_bytes_per_launch": fbytes,
                                    "algorithmic_flop_per_launch": fflop, "traffic": pmc_traffic("fc_chain_bwd_kernel"),
                                    "mfma": {"achieved": ft, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ft / MFMA_F32_PEAK_TFLOPS},
                                    "note": "a dependency chain (4 GEMM stages of 32 rows handed between 8 workgroups inside one "
                                            "launch): bound by hand-off latency, not by either roof"})
            elif lname.startswith("conv_bwd_bx3_kernel<128, 128"):
                longest_out["same_as"] = "roofline"
        heaviest = {"kernel": "sn::conv_bwd_bx3_kernel<128,128,DZ_POOL> (conv5 backward: dgrad + wgrad; fp32 products as "
                              "six bf16 MFMAs of three-way split operands)",
                    "bound": "hbm" if conv_hbm_bound else "mfma",
                    "achieved": conv_gbs if conv_hbm_bound else conv_tf,
                    "peak": HBM_PEAK_GBS if conv_hbm_bound else MFMA_SPLIT_BF16_PEAK_TFLOPS,
                    "unit": "GB/s" if conv_hbm_bound else "TFLOP/s",
                    "frac": conv_gbs / HBM_PEAK_GBS if conv_hbm_bound else conv_tf / MFMA_SPLIT_BF16_PEAK_TFLOPS,
                    "traffic": pmc_traffic("conv_bwd_bx3_kernel<128, 128"),
                    "algorithmic_bytes_per_launch": conv_bytes, "algorithmic_flop_per_launch": conv_flop,
                    "avg_launch_ms": conv_ms,
                    "hbm": hbm_block(conv_gbs),
                    "mfma": {"achieved": conv_tf, "peak": MFMA_SPLIT_BF16_PEAK_TFLOPS, "unit": "fp32-equivalent TFLOP/s",
                             "frac": conv_tf / MFMA_SPLIT_BF16_PEAK_TFLOPS,
                             "vs_fp32_mfma_peak": conv_tf / MFMA_F32_PEAK_TFLOPS},
                    "note": "the heaviest kernel (most bytes and flops of any launch); matrix ceiling = dense bf16 MFMA peak / 6 "
                            "products (tools/micro/bf16x3_gemm.hip: 392 fp32-equivalent TFLOP/s measured, 155 for the fp32 MFMA)"}
        if longest_out is not None and "frac" in longest_out:
            roofline_main = dict(longest_out, role="the longest kernel of the step (largest per-step time in %s)" % prov["dir"])
        elif longest_out is not None and longest_out.get("same_as") == "roofline":
            roofline_main = dict(heaviest, role="the longest kernel of the step per %s = the heaviest one" % prov["dir"])
        else:
            roofline_main = dict(heaviest, role="the heaviest kernel of the step (no live probe for the profile's longest kernel %r)"
                                 % (longest_out or {}).get("kernel"))
        out = {
            "metric": "point-clouds/sec fwd+bwd, Bx1024->64 soft-proj+Chamfer",
            "value": value, "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "timed_steps": total_steps, "timed_region_s": dt, "allreduce_alone_us": allreduce_alone_us,
            "config": {"workload": "BASELINE configs[1]: SampleNet sampler train step (fwd + simplification/projection "
                                   "losses + bwd), B=%d per GPU, 1024->64 points, K=8, bottleneck 128; no optimizer step" % B,
                       "batch_per_gpu": B, "global_batch": B * world, "n_in": N, "n_out": M, "group_size": K,
                       "parallelism": "dp%d" % world, "rccl_ranks": world if reducer.collective else 0,
                       "grad_allreduce": ar if reducer.collective else "none",
                       "execution": "eager launches" if args.no_graph else "whole step replayed as one hipGraph",
                       "mlp": "hand-written MFMA kernels: conv stack = fp32 via split-bf16 products (fp32-accurate), FC head = fp32 MFMA"},
            "profile": prov,
            # `roofline`: the kernel with the LARGEST per-step time in the committed profile (the honest one -- a
            # latency chain far below either roof), measured live; the heaviest kernel and the whole step follow under their own keys
            "roofline": roofline_main,
            "roofline_longest": longest_out,
            # the heaviest GEMM kernel of the step (most flops and most bytes of any launch): backward of the last 1x1 convolution;
            # both roofs, "bound" names the nearer one
            "roofline_heaviest": heaviest,
            # the whole step on SURVEY 8d's algorithmic bytes (launch gaps and the dependency chain are in the denominator)
            "roofline_step": hbm_block(step_algorithmic_bytes(B, N) / (ms * 1e-3) / 1e9, algorithmic_bytes_per_step=step_algorithmic_bytes(B, N),
                                       traffic=pmc_traffic_step() if B == 32 else None, ms_per_step=ms,
                                       note="SURVEY 8d's bytes (3.73 MB per cloud) over the step time; traffic = sum of the committed "
                                            "PMC passes over the step's 14 kernels at B = 32"),
            # the geometric kernel of the path (SURVEY 8d's per-cloud byte count applies to it)
            "roofline_geometry": hbm_block(achieved, kernel="sn::pairscan_kernel<16,true,true> (kNN + soft projection + both Chamfer directions)",
                                           traffic=pmc_traffic("pairscan_kernel<16"), algorithmic_bytes_per_launch=alg, avg_launch_ms=kern_ms,
                                           saturating_batch=hbm_block(sat_gbs, batch=Bsat, clouds_per_s=Bsat / (sat_ms * 1e-3),
                                                                      avg_launch_ms=sat_ms)),
            # the whole step against the fp32 matrix peak: MLP flops per step / step time (the geometric and scalar kernels, the
            # launch gaps and the dependency chain are all in the denominator)
            "step_mfma": {"flop_per_step": step_flop, "achieved": step_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": step_tf / MFMA_F32_PEAK_TFLOPS,
                          "note": "against the fp32 MFMA peak (the conv GEMMs run as split-bf16 products, ceiling %.0f; the FC "
                                  "head on the fp32 MFMA)" % MFMA_SPLIT_BF16_PEAK_TFLOPS},
        }
        def leg(name, fn, *a, **k):
            """A secondary leg must never cost the headline line: its failure is recorded in its place."""
            if args.only_leg and args.only_leg != name:
                return
            import gc

            gc.collect()  # (the previous leg's modules and their captured graphs go now, not during a later capture)
            torch.cuda.synchronize()
            try:
                out[name] = fn(*a, **k)
            except Exception as e:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()

        if world == 1 and not args.no_module_surface:
            leg("module_surface", time_module_surface, dev, B, N, M, K, headline_ms=ms)
        if world == 1 and not args.no_extra_legs:
            leg("config1_classification", time_config1_classification, dev)
            if isinstance(out.get("config1_classification"), dict) and "graph" in out["config1_classification"]:
                out["config1_classification_clouds_per_s"] = out["config1_classification"]["graph"]["value"]
                out["config1_classification_script_clouds_per_s"] = out["config1_classification"]["script"]["value"]
            leg("config3_emd", time_config3_emd, dev)
            leg("config3_sampler", time_config3_sampler, dev)
            if isinstance(out.get("config3_sampler"), dict) and "graph" in out["config3_sampler"]:
                out["config3_sampler_k16_ms"] = out["config3_sampler"]["graph"]["ms_per_step"]
            leg("config5_progressive", time_config5_progressive, dev)
            leg("batch_sweep", time_batch_sweep, dev, N, M, K)
            if isinstance(out.get("batch_sweep"), list):  # (the driver's `parsed` keeps top-level keys only)
                for row in out["batch_sweep"]:
                    if row["batch"] in (512, 2048):
                        out["b%d_clouds_per_s" % row["batch"]] = row["clouds_per_s"]
                        out["b%d_hbm_frac_algorithmic" % row["batch"]] = row["hbm_frac_algorithmic"]
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_reference_model import time_cpu_baseline

            out["cpu_baseline"] = time_cpu_baseline(B, N, M, K, budget_s=args.cpu_budget)
            # the port against the TRUE reference module, timed side by side in the build container (tools/time_reference_cpu.py;
            # /root/reference does not exist on the GPU box) -- throughput ratio port / reference on the same step, same numbers
            try:
                with open(_profile_file("cpu_baseline_reference_vs_port.json")) as f:
                    legs = json.load(f)["legs"]
                out["cpu_baseline"]["port_over_reference_module"] = {k: round(v["port_over_reference_throughput"], 3) for k, v in legs.items()}
                out["cpu_baseline"]["pinned_by"] = "tests/test_oracle.py::test_cpu_baseline_port_matches_reference_run"
            except (OSError, KeyError, ValueError, TypeError):
                pass
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()  # rank 0 was busy with the roofline timing loops: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
