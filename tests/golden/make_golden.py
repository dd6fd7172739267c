#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (itailang/SampleNet, /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

What is executed is the reference's own Python -- registration/src/{soft_projection,samplenet,
sputils}.py imported unmodified, in place, with bytecode writing disabled so nothing is written
under /root/reference -- on CPU.  Three names the reference imports do not exist here and are
supplied by this script:
  * knn_cuda.KNN (third-party wheel, source absent)  -> the (squared distance, index)-ordered
    contract of oracle.knn (see oracle/samplenet_oracle.c: orc_knn);
  * pointnet2.utils.pointnet2_utils.grouping_operation (third-party) -> torch.gather;
  * src.chamfer_distance (JIT-compiles a .cu at import: impossible without CUDA) -> an
    autograd.Function around the reference's OWN compiled CPU functions, oracle/_ref/cd_ref
    (built from chamfer_distance.cpp by oracle/Makefile), mirroring chamfer_distance.py:14-66.
The known-answer tables of the reference's __main__ tests are read out of the reference files
with ast (the numbers are parsed, not retyped):
  registration/src/soft_projection.py:161-222 and classification/soft_projection.py:90-129.
"""
import ast
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SAMPLENET_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- shims
def install_shims():
    knn_cuda = types.ModuleType("knn_cuda")

    class KNN:
        """knn_cuda.KNN(k, transpose_mode=False)(ref (B,C,N), query (B,C,M)) -> dist (B,k,M), idx (B,k,M)."""

        def __init__(self, k, transpose_mode=False):
            self.k, self.t = k, transpose_mode

        def __call__(self, ref, query):
            if not self.t:
                ref, query = ref.permute(0, 2, 1), query.permute(0, 2, 1)
            if ref.dtype == torch.float64:  # the fp64 "exact" run of golden_samplenet_c2: same (distance, index) order
                r, q = ref.detach().numpy(), query.detach().numpy()
                dd = ((q[:, :, None, :] - r[:, None, :, :]) ** 2).sum(-1)  # (B,M,N)
                i = np.argsort(dd, axis=2, kind="stable")[:, :, :self.k]
                d = np.take_along_axis(dd, i, 2)
            else:
                d, i = O.knn(self.k, ref.detach().contiguous().numpy(), query.detach().contiguous().numpy())
            d = torch.from_numpy(np.sqrt(d))
            i = torch.from_numpy(i.astype(np.int64))
            if not self.t:
                d, i = d.permute(0, 2, 1).contiguous(), i.permute(0, 2, 1).contiguous()
            return d, i

    knn_cuda.KNN = KNN
    sys.modules["knn_cuda"] = knn_cuda

    def grouping_operation(features, idx):
        B, C, N = features.shape
        _, M, K = idx.shape
        g = torch.gather(features.unsqueeze(2).expand(B, C, M, N), 3, idx.long().unsqueeze(1).expand(B, C, M, K))
        return g

    p2 = types.ModuleType("pointnet2")
    p2u = types.ModuleType("pointnet2.utils")
    p2uu = types.ModuleType("pointnet2.utils.pointnet2_utils")
    p2uu.grouping_operation = grouping_operation
    p2.utils, p2u.pointnet2_utils = p2u, p2uu
    sys.modules.update({"pointnet2": p2, "pointnet2.utils": p2u, "pointnet2.utils.pointnet2_utils": p2uu})

    cd = O.ref_cd()

    class ChamferDistanceFunction(torch.autograd.Function):
        # mirrors registration/src/chamfer_distance/chamfer_distance.py:14-61 (CPU branch)
        @staticmethod
        def forward(ctx, xyz1, xyz2):
            b, n, _ = xyz1.size()
            m = xyz2.size(1)
            xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
            d1, d2 = torch.zeros(b, n), torch.zeros(b, m)
            i1, i2 = torch.zeros(b, n, dtype=torch.int), torch.zeros(b, m, dtype=torch.int)
            cd.forward(xyz1, xyz2, d1, d2, i1, i2)
            ctx.save_for_backward(xyz1, xyz2, i1, i2)
            ctx.mark_non_differentiable(i1, i2)
            return d1, d2, i1, i2

        @staticmethod
        def backward(ctx, g1, g2, _a, _b):
            xyz1, xyz2, i1, i2 = ctx.saved_tensors
            gx1, gx2 = torch.zeros(xyz1.size()), torch.zeros(xyz2.size())
            cd.backward(xyz1, xyz2, gx1, gx2, g1.contiguous(), g2.contiguous(), i1, i2)
            return gx1, gx2

    class ChamferDistance(torch.nn.Module):
        def forward(self, xyz1, xyz2):
            if xyz1.dtype == torch.float64:  # fp64 "exact" run only: the same minima through plain torch + autograd
                dd = ((xyz1[:, :, None, :] - xyz2[:, None, :, :]) ** 2).sum(-1)
                return dd.min(2)[0], dd.min(1)[0]
            d1, d2, self.last_idx1, self.last_idx2 = ChamferDistanceFunction.apply(xyz1, xyz2)
            return d1, d2

    src = types.ModuleType("src")
    src.__path__ = [os.path.join(REF, "registration", "src")]
    sys.modules["src"] = src
    cdm = types.ModuleType("src.chamfer_distance")
    cdm.ChamferDistance = ChamferDistance
    sys.modules["src.chamfer_distance"] = cdm
    return ChamferDistance


def main_block_arrays(path, names):
    """Evaluate `name = np.array(...)` assignments found in the `if __name__ == '__main__':` block."""
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.If) and "__main__" in ast.dump(node.test):
            for st in ast.walk(node):
                if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
                    nm = st.targets[0].id
                    if nm in names and nm not in out:
                        out[nm] = np.asarray(eval(compile(ast.Expression(st.value), path, "eval"), {"np": np}), dtype=np.float64)
    missing = set(names) - set(out)
    assert not missing, missing
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez(path, **arrs)
    print("wrote", os.path.relpath(path, ROOT), "(%d arrays)" % len(arrs))


# ----------------------------------------------------------------------------- goldens
def golden_known_answers():
    a = main_block_arrays(os.path.join(REF, "registration/src/soft_projection.py"),
                          ["query_cloud", "point_cloud", "point_features", "expected_nn_cloud",
                           "expected_features_nn_1", "expected_features_nn_3"])
    save("known_answer_registration.npz", **a)
    b = main_block_arrays(os.path.join(REF, "classification/soft_projection.py"),
                          ["query_cloud", "point_cloud", "expected_cloud_soft", "expected_cloud_hard"])
    save("known_answer_classification.npz", **b)


def golden_softproj(SoftProjection):
    torch.manual_seed(1234)
    out = {}
    for tag, (B, N, M, K, CF, T) in {"a": (3, 200, 24, 8, 5, 0.7), "b": (2, 64, 9, 3, 4, 0.05), "c": (1, 33, 7, 16, 2, 1.0)}.items():
        P = (torch.rand(B, 3, N) - 0.5).requires_grad_(True)
        Q = (torch.rand(B, 3, M) - 0.5).requires_grad_(True)
        F = torch.randn(B, CF, N, requires_grad=True)
        sp = SoftProjection(K, initial_temperature=T, is_temperature_trainable=True, min_sigma=1e-2)
        proj = sp(P, Q, action="project")
        gp = torch.randn_like(proj)
        gP, gQ, gT = torch.autograd.grad(proj, [P, Q, sp._temperature], gp)
        prop = sp(P, Q, F, action="propagate")
        gprop = torch.randn_like(prop)
        hP, hQ, hF, hT = torch.autograd.grad(prop, [P, Q, F, sp._temperature], gprop)
        proj2, prop2 = sp(P, Q, F, action="project_and_propagate")
        _, idx = sys.modules["knn_cuda"].KNN(K, False)(P.detach(), Q.detach())
        d = dict(P=P, Q=Q, F=F, T=torch.tensor(T), K=torch.tensor(K), idx=idx.permute(0, 2, 1).int(),
                 sigma=sp.sigma(), proj=proj, gproj=gp, gP=gP, gQ=gQ, gT=gT,
                 prop=prop, gprop=gprop, hP=hP, hQ=hQ, hF=hF, hT=hT, proj2=proj2, prop2=prop2)
        out.update({f"{tag}_{k}": v.detach().numpy() for k, v in d.items()})
    save("softproj_reference.npz", **out)


def golden_chamfer(ChamferDistance):
    rng = np.random.default_rng(7)
    out = {}
    for tag, (B, n, m) in {"a": (3, 64, 1024), "b": (2, 100, 37), "c": (1, 1, 5)}.items():
        x1 = torch.from_numpy((rng.random((B, n, 3), dtype=np.float32) - 0.5)).requires_grad_(True)
        x2 = torch.from_numpy((rng.random((B, m, 3), dtype=np.float32) - 0.5))
        if tag == "b":  # exact duplicates: tie-break stress (lowest index must win)
            x2[:, 5] = x2[:, 20]
            x2[:, 6] = x2[:, 20]
            x1 = x1.detach()
            x1[:, 3] = x1[:, 50]
            x1.requires_grad_(True)
        x2.requires_grad_(True)
        cdm = ChamferDistance()
        d1, d2 = cdm(x1, x2)
        g1 = torch.from_numpy(rng.standard_normal(d1.shape).astype(np.float32))
        g2 = torch.from_numpy(rng.standard_normal(d2.shape).astype(np.float32))
        gx1, gx2 = torch.autograd.grad([d1, d2], [x1, x2], [g1, g2])
        d = dict(xyz1=x1, xyz2=x2, dist1=d1, dist2=d2, idx1=cdm.last_idx1, idx2=cdm.last_idx2,
                 gdist1=g1, gdist2=g2, gxyz1=gx1, gxyz2=gx2)
        out.update({f"{tag}_{k}": v.detach().numpy() for k, v in d.items()})
    save("chamfer_reference.npz", **out)


def golden_samplenet(SampleNet):
    """Config C1 (B=4, 1024 -> 64, K=8) train step as registration/main.py:500-531 issues it, plus eval."""
    out = {}
    # "m" shares c1's constructor arguments, hence (same seed) its initial state_dict: only x / outputs / grads are stored
    for tag, (B, N, M, K, bneck, shape) in {"c1": (4, 1024, 64, 8, 128, "bnc"), "s": (3, 96, 12, 5, 32, "bcn"),
                                            "m": (16, 256, 64, 8, 128, "bnc")}.items():
        torch.manual_seed(0)
        net = SampleNet(M, bneck, group_size=K, initial_temperature=1.0, is_temperature_trainable=True,
                        min_sigma=1e-2, input_shape=shape, output_shape=shape)
        # move the BN affine params and temperature off their trivial init so their grads are exercised
        with torch.no_grad():
            for nme, p in net.named_parameters():
                if "bn" in nme:
                    p.add_(0.1 * torch.randn_like(p))
            net.project._temperature.fill_(0.3)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        x = torch.rand(B, N, 3) - 0.5
        if shape == "bcn":
            x = x.permute(0, 2, 1).contiguous()
        net.train()
        simp, proj = net(x)
        x_bnc = x if shape == "bnc" else x.permute(0, 2, 1).contiguous()
        simp_bnc = simp if shape == "bnc" else simp.permute(0, 2, 1).contiguous()
        lsimp = net.get_simplification_loss(x_bnc, simp_bnc, M, 1.0, 0.5 / M)
        lproj = net.get_projection_loss()
        gw = torch.randn_like(proj)
        loss = 0.01 * lsimp + 0.01 * lproj + (proj * gw).sum() / proj.numel()
        loss.backward()
        if tag != "m":
            for k, v in sd0.items():
                out[f"{tag}_sd_{k}"] = v.numpy()
        else:
            assert all(np.array_equal(v.numpy(), out[f"c1_sd_{k}"]) for k, v in sd0.items())
        for k, v in net.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{tag}_sd1_{k}"] = v.numpy()
        for k, p in net.named_parameters():
            out[f"{tag}_grad_{k}"] = p.grad.numpy()
        out.update({f"{tag}_x": x.numpy(), f"{tag}_simp": simp.detach().numpy(), f"{tag}_proj": proj.detach().numpy(),
                    f"{tag}_gw": gw.numpy(), f"{tag}_lsimp": lsimp.detach().numpy(), f"{tag}_lproj": lproj.detach().numpy(),
                    f"{tag}_loss": loss.detach().numpy(),
                    f"{tag}_cfg": np.array([B, N, M, K, bneck, 0 if shape == "bnc" else 1])})
        # eval branch (samplenet.py:119-141) -- its hard-coded .cuda() is neutralised for the CPU run
        net.eval()
        orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            with torch.no_grad():
                simp_e, match = net(x)
        finally:
            torch.Tensor.cuda = orig
        out[f"{tag}_eval_simp"] = simp_e.numpy()
        out[f"{tag}_eval_match"] = match.numpy()
    save("samplenet_reference.npz", **out)


def _c2_case(SampleNet, B, N, M, K, perturb, seed):
    """One C2 step through the reference module in fp32 and fp64 -> (results per precision, fixture arrays of the fp32 run)."""
    res, out = {}, {}
    for prec in ("f32", "f64"):
        torch.manual_seed(0)
        net = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True,
                        min_sigma=1e-2, input_shape="bnc", output_shape="bnc")
        x = torch.rand(B, N, 3) - 0.5
        if perturb:
            torch.manual_seed(100 + seed)
            with torch.no_grad():
                for nme, p in net.named_parameters():
                    if "bn" in nme:
                        p.add_(0.1 * torch.randn_like(p))
                net.project._temperature.fill_(0.3)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        if prec == "f64":
            net = net.double()
            net.project._min_sigma = net.project._min_sigma.double()
            x = x.double()
        net.train()
        simp, proj = net(x)
        lsimp = net.get_simplification_loss(x, simp, M, 1.0, 0.0)
        lproj = net.get_projection_loss()
        loss = 0.01 * lsimp + 0.01 * lproj + proj.mean()
        loss.backward()
        res[prec] = dict(simp=simp, proj=proj, lsimp=lsimp, lproj=lproj, loss=loss,
                         grads={k: p.grad for k, p in net.named_parameters()})
        if prec == "f32":
            for k, v in sd0.items():
                out["sd_" + k] = v.numpy()
            out["x"] = x.numpy()
            out["cfg"] = np.array([B, N, M, K, 128, 0])
            for k, v in net.state_dict().items():
                if "running" in k:
                    out["sd1_" + k] = v.numpy()
    return res, out


def golden_samplenet_c2(SampleNet):
    """Config C2 = BASELINE configs[1], the headline workload (B=32, 1024 -> 64): the sampler's training step exactly as
    bench.py times it -- L = 0.01 * L_simp(gamma=1, delta=0) + 0.01 * L_proj + mean(proj), default torch init under seed 0,
    T = 1 -- for K = 8 (registration default) and K = 7 (classification default, train_samplenet.py:46; BatchNorm affine
    parameters and the temperature moved off their trivial initial values).  Each case is run twice through the reference
    module: in fp32 (the reference as users run it) and in fp64 (same weights and input cast up: the exact answer that both
    the reference's fp32 run and the HIP path approximate; kNN / Chamfer shims switch to plain fp64 numpy / torch there).
    Stored: input, initial state_dict (shared by both cases where equal), simp / proj / losses / every gradient, fp32 + fp64."""
    out = {}
    for tag, (B, N, M, K, perturb) in {"k8": (32, 1024, 64, 8, False), "k7": (32, 1024, 64, 7, True)}.items():
        # A fixture must not sit on a discontinuity of the step (a ReLU input or max-pool pair within rounding of a tie, a
        # neighbour swap): there the reference's OWN fp32 and fp64 runs disagree by ~1e-2 in the gradients and nothing can
        # be pinned.  The perturbation seed is therefore advanced until they agree to 3e-4 (recorded in <tag>_seed).
        for seed in range(16):
            res, tmp = _c2_case(SampleNet, B, N, M, K, perturb, seed)
            g32 = np.concatenate([v.numpy().ravel() for v in res["f32"]["grads"].values()]).astype(np.float64)
            g64 = np.concatenate([v.numpy().ravel() for v in res["f64"]["grads"].values()])
            gap = np.linalg.norm(g32 - g64) / np.linalg.norm(g64)
            print(tag, "seed", seed, "reference fp32 vs fp64 gradient gap %.3g" % gap)
            if gap < 3e-4:
                break
        else:
            raise RuntimeError("no well-conditioned fixture found")
        for k, v in tmp.items():
            if k.startswith("sd_"):
                if tag == "k8" or not np.array_equal(v, out["k8_" + k]):
                    out[f"{tag}_{k}"] = v
            else:
                out[f"{tag}_{k}"] = v
        out[f"{tag}_seed"] = np.array(seed)
        for prec, r in res.items():
            sfx = "" if prec == "f32" else "_f64"
            for k in ("simp", "proj", "lsimp", "lproj", "loss"):
                v = r[k].detach().numpy()
                out[f"{tag}_{k}{sfx}"] = v if v.ndim == 0 else v.astype(np.float32)  # (fp64 scalars stay fp64)
            for k, gr in r["grads"].items():
                out[f"{tag}_grad{sfx}_{k}"] = gr.numpy().astype(np.float32)
        # how far the reference's own fp32 run is from exact arithmetic -- printed for the record (the tests recompute it)
        g32 = np.concatenate([v.numpy().ravel() for v in res["f32"]["grads"].values()]).astype(np.float64)
        g64 = np.concatenate([v.numpy().ravel() for v in res["f64"]["grads"].values()])
        print(tag, "reference fp32 vs fp64: simp max|d| %.3g  loss |d| %.3g  grad rel %.3g" % (
            np.abs(res["f32"]["simp"].detach().numpy() - res["f64"]["simp"].detach().numpy()).max(),
            abs(float(res["f32"]["loss"]) - float(res["f64"]["loss"])), np.linalg.norm(g32 - g64) / np.linalg.norm(g64)))
    np.savez_compressed(os.path.join(HERE, "samplenet_c2_reference.npz"), **out)
    print("wrote tests/golden/samplenet_c2_reference.npz (%d arrays)" % len(out))


def _selection_margin(x, simp, K):
    """Smallest coordinate shift (first order) that would change one of the step's discrete selections on this simplified cloud:
    the K-th / (K+1)-th neighbour of a query, a query's nearest point, the arg-max of the per-cloud maximum term.  fp64."""
    x, q = x.astype(np.float64), simp.astype(np.float64)
    d = ((q[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1)           # (B, M, N)
    ds = np.sort(d, axis=2)
    knn = ((ds[:, :, K] - ds[:, :, K - 1]) / (2 * np.sqrt(ds[:, :, K]) + 1e-12)).min()
    nn = ((ds[:, :, 1] - ds[:, :, 0]) / (2 * np.sqrt(ds[:, :, 1]) + 1e-12)).min()
    m = np.sort(ds[:, :, 0], axis=1)
    amax = ((m[:, -1] - m[:, -2]) / (2 * np.sqrt(m[:, -1]) + 1e-12)).min()
    return float(min(knn, nn, amax)), (float(knn), float(nn), float(amax))


def golden_samplenet_c2_clean(SampleNet):
    """A K = 7 headline fixture WITHOUT near-ties: the k7 case of samplenet_c2_reference.npz sits next to
    neighbour swaps -- a 2e-5 shift of the simplified cloud flips a dozen kNN sets there, and its gradients can then only be
    held to 3e-2.  Here the perturbation seed is searched for the step whose discrete selections (kNN sets, nearest points,
    the arg-max of the maximum term) survive the LARGEST coordinate shift among 600 candidates (recorded as k7c_margin; the HIP
    head sits within 2e-5 of the reference's, typically 5e-6) and whose fp32 / fp64 reference runs agree to 3e-4: that fixture is held to the K = 8 bars
    (no selection flips, gradients 3e-4).  The flipping case stays in the other file as the named discontinuity test."""
    B, N, M, K = 32, 1024, 64, 7
    cand = []
    for seed in range(1000, 1600):
        torch.manual_seed(0)
        net = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True,
                        min_sigma=1e-2, input_shape="bnc", output_shape="bnc")
        x = torch.rand(B, N, 3) - 0.5
        torch.manual_seed(100 + seed)
        with torch.no_grad():
            for nme, p in net.named_parameters():
                if "bn" in nme:
                    p.add_(0.1 * torch.randn_like(p))
            net.project._temperature.fill_(0.3)
            net.train()
            simp, _ = net(x)
        margin, parts = _selection_margin(x.numpy(), simp.numpy(), K)
        cand.append((margin, seed, parts))
    cand.sort(reverse=True)
    print("best selection margins:", [(round(m * 1e5, 2), sd) for m, sd, _ in cand[:8]], "x 1e-5", flush=True)
    for margin, seed, parts in cand[:12]:
        res, tmp = _c2_case(SampleNet, B, N, M, K, True, seed)
        g32 = np.concatenate([v.numpy().ravel() for v in res["f32"]["grads"].values()]).astype(np.float64)
        g64 = np.concatenate([v.numpy().ravel() for v in res["f64"]["grads"].values()])
        gap = np.linalg.norm(g32 - g64) / np.linalg.norm(g64)
        print("seed", seed, "margin %.3g (knn %.3g, nn %.3g, argmax %.3g): reference fp32 vs fp64 gradient gap %.3g" % ((margin,) + parts + (gap,)), flush=True)
        if gap < 3e-4:
            break
    else:
        raise RuntimeError("no tie-free fixture found")
    out = {"k7c_" + k: v for k, v in tmp.items()}
    out["k7c_seed"], out["k7c_margin"] = np.array(seed), np.array(margin)
    for prec, r in res.items():
        sfx = "" if prec == "f32" else "_f64"
        for k in ("simp", "proj", "lsimp", "lproj", "loss"):
            v = r[k].detach().numpy()
            out[f"k7c_{k}{sfx}"] = v if v.ndim == 0 else v.astype(np.float32)
        for k, gr in r["grads"].items():
            out[f"k7c_grad{sfx}_{k}"] = gr.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "samplenet_c2_clean_reference.npz"), **out)
    print("wrote tests/golden/samplenet_c2_clean_reference.npz (%d arrays)" % len(out))


def golden_samplenet_c2_eval(SampleNet):
    """The eval branch (samplenet.py:119-141: KNN(1) + nn_matching with farthest-point completion) at C2 -- B = 32, 1024 -> 64,
    the headline's shapes: default init under seed 0, one training-mode forward to move the running statistics, then the
    module in eval mode; the matched cloud is stored together with how far the nearest / second-nearest input point of every
    generated point are apart (k8e_nn_margin: the coordinate shift that would change a match)."""
    B, N, M, K = 32, 1024, 64, 8
    torch.manual_seed(0)
    net = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-2,
                    input_shape="bnc", output_shape="bnc")
    x = torch.rand(B, N, 3) - 0.5
    net.train()
    with torch.no_grad():
        net(x)
    out = {"k8e_sd_" + k: v.clone().numpy() for k, v in net.state_dict().items()}
    net.eval()
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # (samplenet.py:141 hard-codes .cuda())
    try:
        with torch.no_grad():
            simp, match = net(x)
    finally:
        torch.Tensor.cuda = orig
    d = ((simp.numpy().astype(np.float64)[:, :, None, :] - x.numpy().astype(np.float64)[:, None, :, :]) ** 2).sum(-1)
    ds = np.sort(d, axis=2)
    margin = ((ds[:, :, 1] - ds[:, :, 0]) / (2 * np.sqrt(ds[:, :, 1]) + 1e-12))
    out.update({"k8e_x": x.numpy(), "k8e_eval_simp": simp.numpy(), "k8e_eval_match": match.numpy(),
                "k8e_nn_margin": margin.astype(np.float32), "k8e_cfg": np.array([B, N, M, K, 128, 0])})
    print("C2 eval: smallest nearest-point margin %.3g, clouds whose matches are all distinct: %d / %d" % (
        margin.min(), sum(len(np.unique(d[b].argmin(1))) == M for b in range(B)), B))
    np.savez_compressed(os.path.join(HERE, "samplenet_c2_eval_reference.npz"), **out)
    print("wrote tests/golden/samplenet_c2_eval_reference.npz (%d arrays)" % len(out))


def golden_nn_matching(sputils):
    rng = np.random.default_rng(3)
    B, N, k = 3, 200, 32
    pc = rng.random((B, N, 3), dtype=np.float32)
    idx = rng.integers(0, 40, (B, k))  # many repeats -> exercises unique + FPS completion
    save("nn_matching_reference.npz", pc=pc, idx=idx,
         out_fps=sputils.nn_matching(pc, idx, k, complete_fps=True),
         out_nofps=sputils.nn_matching(pc, idx, k, complete_fps=False))


def golden_pcrnet(ChamferDistance):
    """registration/models/pcrnet.py (PCRNet) + src/quaternion.py (qrot) imported and run as they are; the task loss as
    main.py:557-577 composes it for --loss-type 1 (QuaternionTransform itself needs kornia: its rotate() is the qrot call
    of qdataset.py:106-109, restated here).  Weights: default torch init under a fixed seed -- only their checksums are
    stored (the test rebuilds the same module the same way)."""
    sys.path.insert(0, os.path.join(REF, "registration"))
    pcr = importlib.import_module("models.pcrnet")
    Q = importlib.import_module("src.quaternion")
    torch.manual_seed(21)
    model = pcr.PCRNet(bottleneck_size=256, input_shape="bnc")
    B, N = 3, 160
    g = torch.Generator().manual_seed(22)
    p0 = (torch.rand(B, N, 3, generator=g) - 0.5).requires_grad_(True)
    ang = torch.tensor([0.3, -0.5, 0.8])
    R = torch.stack([torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
                     for a in ang.tolist()])
    p1 = (torch.bmm(p0.detach(), R.transpose(1, 2)) + 0.01 * torch.randn(B, N, 3, generator=g)).contiguous()
    twist, pre = model(p0, p1)
    qnorm = torch.mean((torch.sum(pre ** 2, dim=1) - 1) ** 2)
    quat = twist[:, 0:4].unsqueeze(1).expand([-1, N, -1]).contiguous()
    p1_est = Q.qrot(quat, p0)
    c01, c10 = ChamferDistance()(p1, p1_est)
    loss = torch.mean(c01) + torch.mean(c10)
    loss.backward()
    sums = {("w_" + n.replace(".", "_")): np.float64(p.detach().double().abs().sum()) for n, p in model.named_parameters()}
    gn = {("g_" + n.replace(".", "_")): p.grad.numpy() for n, p in model.named_parameters() if n in ("feat.conv1.weight", "fc6.weight", "fc6.bias")}
    save("pcrnet_reference.npz", p0=p0.detach().numpy(), p1=p1.numpy(), twist=twist.detach().numpy(), pre=pre.detach().numpy(),
         loss=np.float32(loss.item()), qnorm=np.float32(qnorm.item()), grad_p0=p0.grad.numpy(), **sums, **gn)


def golden_samplenet_task(SampleNet, ChamferDistance):
    """The sampler's training step with the REAL task loss, as registration/main.py composes it (compute_samplenet_loss
    :500-537 with NUM_SAMPLED_CLOUDS == 1, compute_pcrnet_loss :557-598 with --loss-type 1):
        p1_simplified, p1_projected = sampler(p1)
        L = pcrnet_chamfer(PCRNet(p0, p1_projected)) + ALPHA * simplification_loss + LMBDA * projection_loss
    through the reference SampleNet, the reference PCRNet (frozen, default init under a fixed seed: the test rebuilds it the same
    way, checksums stored) and the reference's compiled Chamfer, in fp32 and in fp64 (the exact answer).  B = 32, 1024 -> 64,
    K = 8 (BASELINE configs[1] with the registration task on top)."""
    sys.path.insert(0, os.path.join(REF, "registration"))
    pcr_mod = importlib.import_module("models.pcrnet")
    Q = importlib.import_module("src.quaternion")
    B, N, M, K = 32, 1024, 64, 8
    out, res = {}, {}
    for prec in ("f32", "f64"):
        torch.manual_seed(0)
        net = SampleNet(M, 128, group_size=K, initial_temperature=1.0, is_temperature_trainable=True,
                        min_sigma=1e-2, input_shape="bnc", output_shape="bnc")
        p1 = torch.rand(B, N, 3) - 0.5          # source (sampled)
        p0 = torch.rand(B, N, 3) - 0.5          # template (complete)
        torch.manual_seed(31)
        pcr = pcr_mod.PCRNet(bottleneck_size=1024, input_shape="bnc")
        for q in pcr.parameters():
            q.requires_grad_(False)
        if prec == "f32":
            for k, v in net.state_dict().items():
                out["sd_" + k] = v.clone().numpy()
            out["p0"], out["p1"] = p0.numpy(), p1.numpy()
            for n, q in pcr.named_parameters():
                out["pcrsum_" + n.replace(".", "_")] = np.float64(q.detach().double().abs().sum())
        else:
            net, pcr = net.double(), pcr.double()
            net.project._min_sigma = net.project._min_sigma.double()
            p0, p1 = p0.double(), p1.double()
        net.train()
        pcr.eval()
        simp, proj = net(p1)
        lsimp = net.get_simplification_loss(p1, simp, M, 1.0, 0.0)
        lproj = net.get_projection_loss()
        twist, pre = pcr(p0, proj)
        quat = twist[:, 0:4].unsqueeze(1).expand([-1, N, -1]).contiguous()
        p1_est = Q.qrot(quat, p0)
        c01, c10 = ChamferDistance()(proj.contiguous(), p1_est.contiguous())
        task = torch.mean(c01) + torch.mean(c10)
        loss = task + 0.01 * lsimp + 0.01 * lproj
        loss.backward()
        res[prec] = {k: p.grad.numpy() for k, p in net.named_parameters()}
        sfx = "" if prec == "f32" else "_f64"
        for k, v in (("simp", simp), ("proj", proj), ("loss", loss), ("task", task), ("lsimp", lsimp), ("twist", twist)):
            v = v.detach().numpy()
            out[k + sfx] = v if v.ndim == 0 else v.astype(np.float32)
        for k, gr in res[prec].items():
            out["grad%s_%s" % (sfx, k)] = gr.astype(np.float32)
    g32 = np.concatenate([v.ravel() for v in res["f32"].values()]).astype(np.float64)
    g64 = np.concatenate([v.ravel() for v in res["f64"].values()])
    print("task step: reference fp32 vs fp64 gradient gap %.3g, loss %.9f / %.9f" % (
        np.linalg.norm(g32 - g64) / np.linalg.norm(g64), float(out["loss"]), float(out["loss_f64"])))
    np.savez_compressed(os.path.join(HERE, "samplenet_task_reference.npz"), **out)
    print("wrote tests/golden/samplenet_task_reference.npz (%d arrays)" % len(out))


def golden_loaders():
    """Row f4 (reconstruction/src/in_out.py): PLY fixtures WRITTEN by the reference's vendored plyfile package (ascii,
    binary_little_endian, binary_big_endian; vertices + colours + triangle faces) under tests/golden/ply/<syn_id>/<model>.ply
    with what the same package reads back; and the outputs of the reference's own split_data / PointCloudDataSet -- their
    source is lifted out of in_out.py with ast (importing the module would start its dataset download)."""
    import shutil

    sys.path.insert(0, os.path.join(REF, "reconstruction", "external", "python_plyfile"))
    plyfile = importlib.import_module("plyfile")
    root = os.path.join(HERE, "ply")
    shutil.rmtree(root, ignore_errors=True)
    rng = np.random.default_rng(5)
    out = {}
    cases = [("02691156", "a1", True, "<", 12), ("02691156", "b2", False, "<", 12), ("03001627", "c3", False, ">", 12),
             ("03001627", "d4", True, "<", 12), ("04379243", "e5", False, "<", 12)]
    for syn, model, text, order, n in cases:
        os.makedirs(os.path.join(root, syn), exist_ok=True)
        v = np.empty(n, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
        for c in "xyz":
            v[c] = rng.random(n, dtype=np.float32) - 0.5
        for c in ("red", "green", "blue"):
            v[c] = rng.integers(0, 256, n)
        faces = np.empty(5, dtype=[("vertex_indices", "i4", (3,))])
        faces["vertex_indices"] = rng.integers(0, n, (5, 3))
        path = os.path.join(root, syn, model + ".ply")
        plyfile.PlyData([plyfile.PlyElement.describe(v, "vertex"), plyfile.PlyElement.describe(faces, "face")],
                        text=text, byte_order=order).write(path)
        back = plyfile.PlyData.read(path)
        pts = np.vstack([back["vertex"]["x"], back["vertex"]["y"], back["vertex"]["z"]]).T
        out[f"{syn}_{model}_points"] = pts
        out[f"{syn}_{model}_faces"] = np.vstack(back["face"]["vertex_indices"])
        out[f"{syn}_{model}_color"] = np.hstack([np.vstack(back["vertex"][c]) for c in ("red", "green", "blue")])
    # reference split_data / PointCloudDataSet, lifted by ast
    src = open(os.path.join(REF, "reconstruction/src/in_out.py")).read()
    tree = ast.parse(src)
    ns = {"np": np, "range": range, "object": object}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in ("split_data", "PointCloudDataSet"):
            exec(compile(ast.Module([node], []), "in_out.py", "exec"), ns)
    data = rng.random((23, 6, 3), dtype=np.float32)
    labels = np.array(["m%d" % i for i in range(23)], dtype=object)
    tr, va, te, perm = ns["split_data"](data, (0.85, 0.05, 0.10), 42)
    out.update(split_data=data, split_train=tr, split_val=va, split_test=te, split_perm=perm)
    ds = ns["PointCloudDataSet"](data, labels=labels, init_shuffle=False)
    np.random.seed(7)
    ds.shuffle_points(seed=3)
    seq = []
    for _ in range(7):  # crosses two epoch boundaries
        pc, lb, _ = ds.next_batch(8, seed=11)
        seq.append(pc.copy())
    out["ds_batches"] = np.stack(seq)
    out["ds_epochs"] = np.array(ds.epochs_completed)
    fe, fl, _ = ds.full_epoch_data(shuffle=True, seed=13)
    out["ds_full_epoch"] = fe
    out["ds_full_labels"] = np.array([str(v) for v in fl])
    save("loaders_reference.npz", **out)


def golden_modelnet():
    """Row f4 (registration/data/modelnet_loader_torch.py): the reference's ModelNetCls run on a tiny shard set.  h5py is not
    installed here, so the module is imported with a stand-in `h5py` whose File(name) opens the .npz twin of a shard (same two
    arrays, "data" and "label"); everything else -- file lists, concatenation, label shape, per-item point order from numpy's
    global generator, the id2file shape names -- is the reference's own code.  Fixture tree: tests/golden/modelnet/."""
    import shutil
    import types

    root = os.path.join(HERE, "modelnet")
    folder = "modelnet40_ply_hdf5_2048"
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.join(root, folder))
    rng = np.random.default_rng(17)
    shards = {"train": [(5, 2), (4, 2)], "test": [(3, 1)]}  # (clouds, label ndim) per shard
    for split, specs in shards.items():
        names = []
        for i, (n, nd) in enumerate(specs):
            name = "ply_data_%s%d.npz" % (split, i)
            lab = rng.integers(0, 40, (n, 1) if nd == 2 else (n,)).astype(np.uint8)
            np.savez(os.path.join(root, folder, name), data=(rng.random((n, 40, 3), dtype=np.float32) - 0.5), label=lab)
            names.append("data/%s/%s" % (folder, name))
            with open(os.path.join(root, folder, "ply_data_%s_%d_id2file.json" % (split, i)), "w") as f:
                json.dump(["%s/shape_%d_%d.ply" % (split, i, j) for j in range(n)], f)
        with open(os.path.join(root, folder, split + "_files.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
    fake = types.ModuleType("h5py")
    fake.File = lambda name, *a, **k: np.load(name)
    sys.modules["h5py"] = fake
    try:
        spec = importlib.util.spec_from_file_location("ref_modelnet_loader", os.path.join(REF, "registration/data/modelnet_loader_torch.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        del sys.modules["h5py"]
    mod.BASE_DIR = root
    out = {}
    for split, train in (("train", True), ("test", False)):
        ds = mod.ModelNetCls(24, None, train=train, download=False, folder=folder, include_shapes=True)
        np.random.seed(5)
        items = [ds[i] for i in range(len(ds))]
        out[split + "_points"] = np.stack([it[0] for it in items])
        out[split + "_labels"] = np.stack([it[1].numpy() for it in items])
        out[split + "_shapes"] = np.array([it[2] for it in items])
        out[split + "_all_points"], out[split + "_all_labels"] = ds.points, ds.labels
    save("modelnet_reference.npz", **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference checkout not found: " + REF
    O.build(ref=True)
    ChamferDistance = install_shims()
    sp_mod = importlib.import_module("src.soft_projection")
    sn_mod = importlib.import_module("src.samplenet")
    sputils = importlib.import_module("src.sputils")
    jobs = {"known": golden_known_answers, "softproj": lambda: golden_softproj(sp_mod.SoftProjection),
            "chamfer": lambda: golden_chamfer(ChamferDistance), "samplenet": lambda: golden_samplenet(sn_mod.SampleNet),
            "c2": lambda: golden_samplenet_c2(sn_mod.SampleNet), "c2clean": lambda: golden_samplenet_c2_clean(sn_mod.SampleNet),
            "c2eval": lambda: golden_samplenet_c2_eval(sn_mod.SampleNet), "nn_matching": lambda: golden_nn_matching(sputils),
            "pcrnet": lambda: golden_pcrnet(ChamferDistance), "loaders": golden_loaders, "modelnet": golden_modelnet,
            "task": lambda: golden_samplenet_task(sn_mod.SampleNet, ChamferDistance)}
    for name in (sys.argv[1:] or list(jobs)):  # python make_golden.py [job ...]   (default: all)
        jobs[name]()
    left = [p for p, _, fs in os.walk(REF) for f in fs if f.endswith(".pyc") or f == "__pycache__"]
    assert not left, "bytecode leaked into the reference tree: %s" % left[:3]
