#!/usr/bin/env python3
"""python tools/graph_trace.py task|module|cls|c5 [replays]: capture one step as a hipGraph, replay it, for
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/graph_trace.py task 30
then  python tools/graph_trace.py --analyze DIR/t_kernel_trace.csv  prints the LAST replay as a timeline: per kernel start offset,
duration and the gap in front of it (what the dependent launches of a latency-bound step are made of).
  task   -- frozen PCRNet on (template 1024 pts, 64 projected pts) + Chamfer, forward + gradient to the projected points
  module -- sampler step + that task term (bench.py module_surface.graph)
  cls    -- the classification sampler's step (bench.py config1_classification.graph)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def analyze(path):
    import csv

    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last replay: walk back from the end until the first kernel name of the sequence repeats
    names = [r["Kernel_Name"] for r in rows]
    # period detection: smallest p such that the last 2p names are two equal halves
    n = len(names)
    # one replay = the stretch between two occurrences of a kernel that runs ONCE per step: the rarest name of the trace's tail (its
    # last two occurrences delimit the last complete replay; kernels of forked branches may interleave differently from replay to
    # replay, so comparing name sequences for an exact period is not robust)
    from collections import Counter

    tail = names[-min(n, 4000):]
    cnt = Counter(tail)
    # (most kernel NAMES of a step occur once per step: the most common occurrence count among the names is the number of steps in
    #  the tail; a name with exactly that count is a once-per-step kernel -- names seen only during warm-up / capture do not qualify)
    common = Counter(v for v in cnt.values() if v >= 3).most_common(1)
    steps_in_tail = common[0][0] if common else 0
    marker = min((k for k, v in cnt.items() if v == steps_in_tail), default=None)
    per = None
    if marker is not None:
        occ = [i for i, nm in enumerate(names) if nm == marker]
        if len(occ) >= 3:
            per = occ[-1] - occ[-2]
            n = occ[-1]  # the last COMPLETE replay ends where the marker starts again
            rows, names = rows[:n], names[:n]
    if per is None:
        per = min(n, 60)
    last = rows[n - per:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    tot_k = 0
    print("%-90s %9s %8s %8s" % ("kernel", "start us", "dur us", "gap us"))
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        tot_k += e - s
        print("%-90s %9.2f %8.2f %8.2f" % (r["Kernel_Name"][:90], (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
    span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
    print("kernels %d  span %.1f us  sum of kernels %.1f us  gaps %.1f us" % (per, span, tot_k / 1e3, span - tot_k / 1e3))
    # period between consecutive replays (start to start)
    if n >= 2 * per:
        print("replay period %.1f us" % ((int(rows[n - per]["Start_Timestamp"]) - int(rows[n - 2 * per]["Start_Timestamp"])) / 1e3))


if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    analyze(sys.argv[2])
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "task"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
B, N, M, K = 32, 1024, 64, 8
if mode == "task":
    from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss

    torch.manual_seed(0)
    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").to(dev).eval()
    for p in pcr.parameters():
        p.requires_grad_(False)
    pcr.static_weights()
    g = torch.Generator(device=dev).manual_seed(5)
    template = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    q = (torch.rand(B, M, 3, device=dev, generator=g) - 0.5).requires_grad_(True)

    def step():
        q.grad = None
        t = pcrnet_chamfer_loss(pcr, template, q)[0]
        t.backward()
        return t

    ms, _ = bench._graph_replay_ms(step, reps=reps)
    print("task_only ms", ms)
elif mode == "module":
    from samplenet_amd import SampleNet
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer
    from samplenet_amd.task_features import PCRNet, pcrnet_chamfer_loss

    torch.manual_seed(0)
    net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc").to(dev).train()
    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc").to(dev).eval()
    for p in pcr.parameters():
        p.requires_grad_(False)
    pcr.static_weights()
    x = torch.rand(B, N, 3, device=dev) - 0.5
    template = torch.rand(B, N, 3, device=dev) - 0.5
    st = SamplerTrainStep(net, x, alpha=0.01, lmbda=0.01, task_loss=lambda p: pcrnet_chamfer_loss(pcr, template, p)[0],
                          reducer=FlatGradAllReducer(net), use_graph=True, input_ring=[x])
    for _ in range(reps):
        st.replay(0)
    torch.cuda.synchronize()
    print("module step done")
elif mode == "c5":
    out = bench.time_config5_progressive(dev, steps=5)
    print({k: v for k, v in out.items() if isinstance(v, dict)})
else:
    out = bench.time_config1_classification(dev, steps=reps)
    print(out)
