"""Times the TRUE reference module (registration/src/samplenet.py imported from /root/reference, as tests/golden/make_golden.py
imports it) beside bench.py's `cpu_baseline` port (oracle/cpu_reference_model.SampleNetCPU) on this container's host cores, on the
unit of work of SURVEY.md 8d (B = 32, 1024 -> 64, K = 8; forward + both losses + mean(proj) + backward), and writes the ratio to
profiles/r04/cpu_baseline_reference_vs_port.json.  Runs only where /root/reference exists (the build container); the GPU box's
bench leg times the port, whose arithmetic tests/test_oracle.py::test_cpu_baseline_port_matches_reference_run pins to a reference run.

The reference needs two CUDA-only third-party packages; the stand-ins are the ones BASELINE.md / SURVEY 8d prescribe:
kNN = broadcast squared distance + topk (knn_cuda.KNN), grouping = torch.gather (pointnet2 grouping_operation); Chamfer is the
reference's own compiled CPU loop (oracle/_ref/cd_ref, built from chamfer_distance.cpp).

    python tools/time_reference_cpu.py [seconds-per-leg]
"""
import importlib
import json
import os
import sys
import time
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SAMPLENET_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from oracle.cpu_reference_model import SampleNetCPU, host_cpu_budget  # noqa: E402


def install_shims():
    knn_cuda = types.ModuleType("knn_cuda")

    class KNN:  # knn_cuda.KNN(k, transpose_mode=False)(ref (B,C,N), query (B,C,M)) -> dist (B,k,M), idx (B,k,M)
        def __init__(self, k, transpose_mode=False):
            self.k, self.t = k, transpose_mode

        def __call__(self, ref, query):
            if self.t:
                ref, query = ref.permute(0, 2, 1), query.permute(0, 2, 1)
            d = ((query.unsqueeze(3) - ref.unsqueeze(2)) ** 2).sum(1)  # (B, M, N)
            dist, idx = d.topk(self.k, dim=2, largest=False)
            dist, idx = dist.sqrt().permute(0, 2, 1).contiguous(), idx.permute(0, 2, 1).contiguous()
            if self.t:
                dist, idx = dist.permute(0, 2, 1).contiguous(), idx.permute(0, 2, 1).contiguous()
            return dist, idx

    knn_cuda.KNN = KNN
    sys.modules["knn_cuda"] = knn_cuda

    def grouping_operation(features, idx):
        B, C, N = features.shape
        _, M, K = idx.shape
        return torch.gather(features.unsqueeze(2).expand(B, C, M, N), 3, idx.long().unsqueeze(1).expand(B, C, M, K))

    p2, p2u, p2uu = types.ModuleType("pointnet2"), types.ModuleType("pointnet2.utils"), types.ModuleType("pointnet2.utils.pointnet2_utils")
    p2uu.grouping_operation = grouping_operation
    p2.utils, p2u.pointnet2_utils = p2u, p2uu
    sys.modules.update({"pointnet2": p2, "pointnet2.utils": p2u, "pointnet2.utils.pointnet2_utils": p2uu})
    cd = O.ref_cd()

    class ChamferDistanceFunction(torch.autograd.Function):  # the CPU branch of chamfer_distance.py:14-61
        @staticmethod
        def forward(ctx, xyz1, xyz2):
            b, n, _ = xyz1.size()
            m = xyz2.size(1)
            xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
            d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
            i1, i2 = torch.zeros(b, n, dtype=torch.int), torch.zeros(b, m, dtype=torch.int)
            cd.forward(xyz1, xyz2, d1, d2, i1, i2)
            ctx.save_for_backward(xyz1, xyz2, i1, i2)
            return d1, d2

        @staticmethod
        def backward(ctx, g1, g2):
            xyz1, xyz2, i1, i2 = ctx.saved_tensors
            gx1, gx2 = torch.zeros(xyz1.size()), torch.zeros(xyz2.size())
            cd.backward(xyz1, xyz2, gx1, gx2, g1.contiguous(), g2.contiguous(), i1, i2)
            return gx1, gx2

    class ChamferDistance(torch.nn.Module):
        def forward(self, xyz1, xyz2):
            return ChamferDistanceFunction.apply(xyz1, xyz2)

    src = types.ModuleType("src")
    src.__path__ = [os.path.join(REF, "registration", "src")]
    sys.modules["src"] = src
    cdm = types.ModuleType("src.chamfer_distance")
    cdm.ChamferDistance = ChamferDistance
    sys.modules["src.chamfer_distance"] = cdm


def timed(step, budget_s):
    for _ in range(3):
        step()
    ts = []
    end = time.perf_counter() + budget_s
    while time.perf_counter() < end or len(ts) < 3:
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def main():
    assert os.path.isdir(REF), "reference checkout not found: " + REF
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    O.build(ref=True)
    install_shims()
    SampleNet = importlib.import_module("src.samplenet").SampleNet
    B, N, M, K = 32, 1024, 64, 8
    out = {"workload": "B=%d, %d->%d, K=%d: forward + 0.01 L_simp + 0.01 L_proj + mean(proj), backward (SURVEY.md 8d)" % (B, N, M, K),
           "host_cpus_granted": host_cpu_budget(), "torch": torch.__version__, "legs": {}}
    for threads in sorted({max(1, host_cpu_budget() // 2), host_cpu_budget()}):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        ref = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-2,
                        input_shape="bnc", output_shape="bnc").train()
        port = SampleNetCPU(M, 128, K).train()
        port.load_state_dict(ref.state_dict(), strict=True)
        x = torch.rand(B, N, 3) - 0.5

        def ref_step():
            for p in ref.parameters():
                p.grad = None
            simp, proj = ref(x)
            loss = 0.01 * ref.get_simplification_loss(x, simp, M, 1, 0) + 0.01 * ref.get_projection_loss() + proj.mean()
            loss.backward()
            return loss

        def port_step():
            for p in port.parameters():
                p.grad = None
            simp, proj = port(x)
            loss = 0.01 * port.get_simplification_loss(x, simp, M, 1, 0) + 0.01 * port.sigma() + proj.mean()
            loss.backward()
            return loss

        lr, lp = float(ref_step().detach()), float(port_step().detach())
        assert abs(lr - lp) <= 1e-6 * max(1.0, abs(lr)), (lr, lp)  # same step, same numbers
        tr, nr = timed(ref_step, budget)
        tp, np_ = timed(port_step, budget)
        out["legs"]["threads_%d" % threads] = {
            "reference_module": {"ms_per_step": tr * 1e3, "clouds_per_s": B / tr, "steps": nr},
            "port": {"ms_per_step": tp * 1e3, "clouds_per_s": B / tp, "steps": np_},
            "port_over_reference_throughput": tr / tp, "loss_reference": lr, "loss_port": lp}
        print(threads, out["legs"]["threads_%d" % threads])
    left = [p for p, _, fs in os.walk(REF) for f in fs if f.endswith(".pyc")]
    assert not left, "bytecode leaked into the reference tree"
    path = os.path.join(ROOT, "profiles", "r04", "cpu_baseline_reference_vs_port.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
