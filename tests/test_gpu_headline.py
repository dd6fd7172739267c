"""The headline workload (BASELINE configs[1] = SURVEY C2: B=32, 1024 -> 64, K=8 and K=7) against the REFERENCE module.

tests/golden/samplenet_c2_reference.npz holds the sampler's training step exactly as bench.py times it
    L = 0.01 * L_simp(gamma=1, delta=0) + 0.01 * L_proj + mean(proj)
run through registration/src/samplenet.py on CPU twice: in fp32 (the reference as it is used) and in fp64 (same weights
and input cast up -- the exact answer both fp32 implementations approximate).  Here the same step runs through the path
bench.py measures -- engine.SamplerTrainStep on an input ring: fused_step.SamplerStepFunction (fc4 inside the pair scan,
per-point minima as atomically combined keys, loss tail inside the conv backward's closing kernel), gradients written
into the flat all-reduce bucket, replayed as a hipGraph -- and, for comparison, through the plain module surface.

Bars.  Integer / index work is bit-exact against the oracle on the simplified cloud the step produced; the geometric
floating-point results on that cloud are within 1e-6.  Against the reference RUN the simplified cloud itself differs
by the summation order of ~450 fp32 dot products per output through 8 BatchNorms (the reference's own fp32 run sits
1.5e-5 from exact arithmetic), so every comparison is made twice: against the reference's fp32 outputs with a fixed bar,
and against the fp64 outputs with the bar "no further from exact than a small multiple of the reference's own fp32 error".
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ALPHA, LMBDA, GAMMA, DELTA = 0.01, 0.01, 1.0, 0.0


def _net(g, tag):
    from samplenet_amd import SampleNet

    B, N, M, K, bneck, _ = [int(v) for v in g[f"{tag}_cfg"]]
    net = SampleNet(M, bneck, group_size=K, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-2,
                    input_shape="bnc", output_shape="bnc")
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("k8_sd_")}
    sd.update({k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_sd_")})  # entries that differ
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return net.cuda().train(), (B, N, M, K)


def _fixture(golden, tag):
    """k8 / k7: samplenet_c2_reference.npz.  k7c: samplenet_c2_clean_reference.npz -- K = 7 WITHOUT near-ties (the generator picked,
    among 600 perturbation seeds, the step whose kNN sets / nearest points / arg-max survive the largest coordinate shift): held
    to the K = 8 bars, no selection flips allowed; k7 stays as the named discontinuity case (flips tolerated, gradients 3e-2)."""
    return golden("samplenet_c2_clean_reference.npz" if tag == "k7c" else "samplenet_c2_reference.npz")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b)), float(np.linalg.norm(b))


def _selection_flips(oracle, xn, simp_a, simp_b, K):
    """How many discrete selections differ between two simplified clouds (B,M,3) against the same input cloud."""
    _, ka = oracle.knn(K, xn, np.ascontiguousarray(simp_a))
    _, kb = oracle.knn(K, xn, np.ascontiguousarray(simp_b))
    a1, ai1, a2, ai2 = oracle.chamfer_forward(np.ascontiguousarray(simp_a), xn)
    b1, bi1, b2, bi2 = oracle.chamfer_forward(np.ascontiguousarray(simp_b), xn)
    return {"knn": int((np.sort(ka, 2) != np.sort(kb, 2)).any(2).sum()), "idx1": int((ai1 != bi1).sum()),
            "idx2": int((ai2 != bi2).sum()), "argmax": int((a1.argmax(1) != b1.argmax(1)).sum())}


def _check_against_reference(g, tag, oracle, x, y_bcn, proj_bmc, loss, lsimp, grads, K, M, sigma):
    """Shared assertions: (simp, proj, loss, every gradient) of one HIP step vs the golden fp32 / fp64 reference runs."""
    simp = y_bcn.permute(0, 2, 1).contiguous().cpu().numpy()  # (B,M,3) as the reference returns it
    ref32, ref64 = g[f"{tag}_simp"], g[f"{tag}_simp_f64"]
    ref_err = np.abs(ref32.astype(np.float64) - ref64).max()
    e_simp32, e_simp64 = np.abs(simp - ref32).max(), np.abs(simp - ref64).max()
    print("\n[%s] simp max|d|: vs ref fp32 %.2e, vs fp64 %.2e (reference fp32 vs fp64 %.2e)" % (tag, e_simp32, e_simp64, ref_err))
    print("[%s] loss %.9f  ref fp32 %.9f  fp64 %.9f" % (tag, float(loss), float(g[f"{tag}_loss"]), float(g[f"{tag}_loss_f64"])))
    if lsimp is not None:
        print("[%s] lsimp %.8f  ref fp32 %.8f  fp64 %.8f" % (tag, float(lsimp), float(g[f"{tag}_lsimp"]), float(g[f"{tag}_lsimp_f64"])))
    # measured 2.0-2.6e-5 / 1.36-1.61e-5 over the three fixtures and four routes: the HIP head is as close to exact arithmetic as
    # the reference's own fp32 (MKL) run is (1.5e-5) -- north_star's 1e-5 sits below the distance between two correct fp32
    # evaluations of this head, so the bars are: within 10 % of the reference's own error against the fp64 run, and within the sum
    # of the two fp32 errors of the reference's fp32 run
    assert e_simp32 <= 3e-5 and e_simp64 <= 1.1 * ref_err + 1e-6
    # discrete selections of the geometry (kNN sets, both Chamfer argmins, the arg-max of the max term) on OUR simplified cloud
    # vs on the reference's: a 1e-5 shift of a query flips a handful of near-ties; each flip re-routes a gradient contribution
    # (the max term carries 1/B of the loss on ONE point), which is what bounds the gradient agreement below
    xn = x.cpu().numpy()
    flips = _selection_flips(oracle, xn, simp, ref32, K)
    print("[%s] selection flips vs the reference's simplified cloud: %s" % (tag, flips))
    assert flips["knn"] <= 12 and flips["idx1"] <= 2 and flips["idx2"] <= 40 and flips["argmax"] <= 1
    if tag != "k7":  # the tie-free fixtures: no kNN set, nearest point or arg-max may differ from the reference's
        assert flips["knn"] == 0 and flips["idx1"] == 0 and flips["argmax"] == 0, flips
    # the geometry of THIS simplified cloud, restated by the oracle: indices bit-exact, projection / loss 1e-6
    _, oidx = oracle.knn(K, xn, simp)
    oproj, _, _ = oracle.softproj_forward(xn.transpose(0, 2, 1), simp.transpose(0, 2, 1), oidx, sigma)
    np.testing.assert_allclose(proj_bmc.cpu().numpy(), oproj.transpose(0, 2, 1), rtol=0, atol=1e-6)
    od1, _, od2, _ = oracle.chamfer_forward(simp, xn)
    olsimp = od1.mean(dtype=np.float64) + od1.max(1).mean(dtype=np.float64) + (GAMMA + DELTA * M) * od2.mean(dtype=np.float64)
    oloss = ALPHA * olsimp + LMBDA * sigma + oproj.mean(dtype=np.float64)
    assert abs(float(loss) - oloss) <= 1e-6 * max(1.0, abs(oloss))
    if lsimp is not None:
        assert abs(float(lsimp) - olsimp) <= 1e-6 * max(1.0, abs(olsimp))
        assert abs(float(lsimp) - float(g[f"{tag}_lsimp"])) <= 1e-5 * float(g[f"{tag}_lsimp"])
    # against the reference run: north_star's 1e-5 on the loss
    assert abs(float(loss) - float(g[f"{tag}_loss"])) <= 1e-5 * max(1.0, abs(float(g[f"{tag}_loss"])))
    assert abs(float(loss) - float(g[f"{tag}_loss_f64"])) <= 1e-5
    # projected cloud against the reference run: a 1e-5 shift of a query can swap its K-th / (K+1)-th neighbour, where the
    # projection is discontinuous -- all but a handful of the 6144 coordinates agree to 1e-4
    close = np.isclose(proj_bmc.cpu().numpy(), g[f"{tag}_proj"], rtol=0, atol=1e-4)
    assert close.mean() >= 0.995, close.mean()
    # gradients: per tensor, relative to its norm.  Biases in front of a BatchNorm have zero true gradient (1e-15 in fp64):
    # both sides hold rounding noise there.
    gmax = max(np.linalg.norm(g[k].astype(np.float64)) for k in g.files if k.startswith(f"{tag}_grad_f64_"))
    bad, worst = [], (0.0, "")
    for name, got in grads.items():
        r32 = g[f"{tag}_grad_{name}"].astype(np.float64)
        r64 = g[f"{tag}_grad_f64_{name}"].astype(np.float64)
        got = got.detach().cpu().numpy().astype(np.float64)
        n64 = np.linalg.norm(r64)
        if n64 < 1e-9 * gmax:
            if np.linalg.norm(got) > 1e-5 * gmax:
                bad.append((name, "zero-gradient bias", np.linalg.norm(got)))
            continue
        e32, e64 = np.linalg.norm(got - r32), np.linalg.norm(got - r64)
        eref = np.linalg.norm(r32 - r64)  # the reference's own fp32 error
        worst = max(worst, (e64 / n64, name))
        # fixed bar against the fp32 reference run.  No selection flips (K=8 fixture): measured 1-3e-5 on the bench path, 9e-5
        # through the module surface -> 3e-4.  With flips (K=7 fixture: the arg-max of the max term or a Chamfer argmin lands
        # on another point; the REFERENCE's own fp32 and fp64 runs differ by 5e-3 for the same reason): measured 1.4e-2 -> 3e-2.
        clean = flips["knn"] + flips["idx1"] + flips["argmax"] == 0  # (an idx2 flip moves 1/(B*N) of the loss: negligible)
        bar32 = 3e-4 if clean else 3e-2
        print("[%s] grad %-22s vs ref fp32 %.2e  vs fp64 %.2e  (reference fp32 vs fp64 %.2e)" % (tag, name, e32 / n64, e64 / n64, eref / n64))
        if e32 > bar32 * n64 or e64 > max(4 * eref, bar32 * n64):
            bad.append((name, e32 / n64, e64 / n64, eref / n64))
    assert not bad, bad
    return worst


@pytest.mark.parametrize("tag", ["k8", "k7", "k7c"])
def test_bench_path_matches_reference_c2(golden, oracle, tag):
    """bench.py's own execution path (graph replay of the fused step on an input ring, gradients in the flat bucket)."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    g = _fixture(golden, tag)
    net, (B, N, M, K) = _net(g, tag)
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    ring = [x.clone(), (torch.rand_like(x) - 0.5)]
    red = FlatGradAllReducer(net)
    step = SamplerTrainStep(net, ring[0], alpha=ALPHA, lmbda=LMBDA, gamma=GAMMA, delta=DELTA, reducer=red, use_graph=True,
                            input_ring=ring)
    assert step._fast_path() and step._ring_graphs, "this test must exercise the path bench.py times"
    step.replay(1)           # another batch in between: the replay of entry 0 must not depend on what ran before
    loss = step.replay(0)
    torch.cuda.synchronize()
    y, proj = step.outputs
    grads = {n: p.grad for n, p in net.named_parameters()}
    assert all(gr is not None and gr.untyped_storage().data_ptr() == red.flat.untyped_storage().data_ptr() for gr in grads.values())
    sigma = float(net.project.sigma())
    worst = _check_against_reference(g, tag, oracle, x, y, proj, loss, None, grads, K, M, sigma)
    print("bench path %s: worst gradient error vs fp64 %.2e (%s)" % (tag, worst[0], worst[1]))
    # BatchNorm running statistics: warm-up (3) + capture (0 executions) + 2 replays moved them; one more step from the golden's
    # state is checked by the module-surface test below


@pytest.mark.parametrize("route", ["op_by_op", "captured"])
@pytest.mark.parametrize("tag", ["k8", "k7", "k7c"])
def test_module_surface_matches_reference_c2(golden, oracle, tag, route, monkeypatch):
    """The same step through the drop-in module surface (forward + get_simplification_loss + get_projection_loss +
    autograd), as registration/main.py:507-531 issues it.  route "captured": those calls replay the two graphs of
    samplenet_amd/surface.py (captured at the first call here instead of after two warm steps, so that exactly one training step
    has run from the fixture's state -- the capture's own warm-up pass restores the running statistics it advanced)."""
    from samplenet_amd import surface

    g = _fixture(golden, tag)
    net, (B, N, M, K) = _net(g, tag)
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    if route == "captured":
        monkeypatch.setattr(surface, "WARM_STEPS", 0)
    else:
        net.graph_surface = False
    simp, proj = net(x)
    assert bool(surface.plans(net)) == (route == "captured")
    lsimp = net.get_simplification_loss(x, simp, M, GAMMA, DELTA)
    lproj = net.get_projection_loss()
    loss = ALPHA * lsimp + LMBDA * lproj + proj.mean()
    loss.backward()
    assert abs(float(lproj.detach()) - float(g[f"{tag}_lproj"])) <= 1e-7
    grads = {n: p.grad for n, p in net.named_parameters()}
    _check_against_reference(g, tag, oracle, x, simp.detach().permute(0, 2, 1), proj.detach(), loss.detach(), lsimp.detach(),
                             grads, K, M, float(lproj.detach()))
    for k in g.files:  # BatchNorm running statistics after exactly one training step
        if k.startswith(f"{tag}_sd1_"):
            np.testing.assert_allclose(net.state_dict()[k[len(tag) + 5:]].cpu().numpy(), g[k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["k8", "k7", "k7c"])
def test_external_task_path_matches_reference_c2(golden, oracle, tag):
    """The fused step with the task loss OUTSIDE the node (engine fast path, task_loss given: proj is a differentiable output
    and the task gradient re-enters the loss backward as an explicit tensor -- sn_sampler_step_loss_keys(grad_proj)) on the same
    fixture: the task is the headline's own mean(proj), so every bar of the bench-path test applies unchanged."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer

    g = _fixture(golden, tag)
    net, (B, N, M, K) = _net(g, tag)
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    ring = [x.clone(), (torch.rand_like(x) - 0.5)]
    red = FlatGradAllReducer(net)
    step = SamplerTrainStep(net, ring[0], alpha=ALPHA, lmbda=LMBDA, gamma=GAMMA, delta=DELTA, reducer=red, use_graph=True,
                            input_ring=ring, task_loss=lambda p: p.mean())
    assert step._fast_path() and step._ring_graphs
    step.replay(1)
    loss = step.replay(0)
    torch.cuda.synchronize()
    step.check()
    y, proj = step.outputs
    grads = {n: p.grad for n, p in net.named_parameters()}
    sigma = float(net.project.sigma())
    worst = _check_against_reference(g, tag, oracle, x, y, proj, loss, None, grads, K, M, sigma)
    print("external-task path %s: worst gradient error vs fp64 %.2e (%s)" % (tag, worst[0], worst[1]))


def _task_fixture(g):
    from samplenet_amd import SampleNet
    from samplenet_amd.task_features import PCRNet

    net = SampleNet(64, 128, group_size=8, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-2,
                    input_shape="bnc", output_shape="bnc")
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    torch.manual_seed(31)  # the generator built the reference PCRNet under this seed (checksums below)
    pcr = PCRNet(bottleneck_size=1024, input_shape="bnc")
    for n, q in pcr.named_parameters():
        assert abs(float(q.detach().double().abs().sum()) - float(g["pcrsum_" + n.replace(".", "_")])) <= 1e-9 * float(q.numel()), n
        q.requires_grad_(False)
    return net.cuda().train(), pcr.cuda().eval(), torch.from_numpy(g["p0"]).cuda(), torch.from_numpy(g["p1"]).cuda()


def test_script_with_task_network_on_captured_surfaces_matches_reference_run(golden, monkeypatch):
    """registration/main.py:500-537 + 557-598 as a SCRIPT issues them -- net(x), the two loss getters, the frozen task network's
    Chamfer term on the projected points, backward() -- with every call on captured work (surface.py for the sampler,
    graphed.py for the task network), against the reference run, under the bars of test_task_step_matches_reference_run."""
    from samplenet_amd import graphed, surface
    from samplenet_amd.task_features import pcrnet_chamfer_loss

    monkeypatch.setattr(surface, "WARM_STEPS", 0)
    monkeypatch.setattr(graphed, "WARM_STEPS", 0)
    g = golden("samplenet_task_reference.npz")
    net, pcr, p0, p1 = _task_fixture(g)
    simp, proj = net(p1)
    loss = pcrnet_chamfer_loss(pcr, p0, proj)[0] + ALPHA * net.get_simplification_loss(p1, simp, 64, GAMMA, DELTA) \
        + LMBDA * net.get_projection_loss()
    loss.backward()
    torch.cuda.synchronize()
    net.check()
    assert bool(surface.plans(net))
    assert any(isinstance(p, graphed._Plan) for p in pcr.__dict__["_sn_graphed"].values())
    _check_task_step(g, "script", loss.detach(), simp.detach(), proj.detach(), net)


@pytest.mark.parametrize("path", ["fused_graph", "fused_eager", "general"])
def test_task_step_matches_reference_run(golden, path):
    """registration/main.py:500-537 + 557-598 (--loss-type 1, one sampled cloud) END TO END against the reference run
    (tests/golden/samplenet_task_reference.npz: reference SampleNet + reference PCRNet + reference Chamfer, fp32 and fp64):
        L = Chamfer(proj, rotate(p0 by PCRNet(p0, proj))) + 0.01 * L_simp + 0.01 * L_proj
    through engine.SamplerTrainStep(task_loss=...) -- the fused single-node step with the task loss outside the node, captured
    and eager, and the op-by-op general path.  Bars as the headline test: loss 1e-5 (north_star), simplified cloud 3e-5, every
    sampler gradient relative to its norm against the reference's fp32 run (3e-3) and no further from the fp64 run than
    4 x the reference's own fp32 error (or the same fixed bar)."""
    from samplenet_amd.engine import SamplerTrainStep
    from samplenet_amd.parallel import FlatGradAllReducer
    from samplenet_amd.task_features import pcrnet_chamfer_loss

    g = golden("samplenet_task_reference.npz")
    net, pcr, p0, p1 = _task_fixture(g)
    red = FlatGradAllReducer(net)
    step = SamplerTrainStep(net, p1, alpha=ALPHA, lmbda=LMBDA, gamma=GAMMA, delta=DELTA, reducer=red,
                            use_graph=(path == "fused_graph"), fused_loss=(path != "general"),
                            task_loss=lambda proj: pcrnet_chamfer_loss(pcr, p0, proj)[0])
    assert step._fast_path() == (path != "general")
    loss = step(p1)
    torch.cuda.synchronize()
    step.check()
    y, proj = step.outputs
    simp = y.permute(0, 2, 1) if path != "general" else y  # fast path: (B,3,M); general path: the module's 'bnc' output
    _check_task_step(g, path, loss, simp, proj, net)


def _check_task_step(g, path, loss, simp, proj, net):
    e_simp = float((simp.cpu() - torch.from_numpy(g["simp"])).abs().max())
    print("\n[%s] loss %.9f  ref fp32 %.9f  fp64 %.9f   simp max|d| %.2e" % (path, float(loss), float(g["loss"]), float(g["loss_f64"]), e_simp))
    assert e_simp <= 3e-5
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 and abs(float(loss) - float(g["loss_f64"])) <= 1e-5
    close = np.isclose(proj.cpu().numpy(), g["proj"], rtol=0, atol=1e-4)
    assert close.mean() >= 0.995, close.mean()
    gmax = max(np.linalg.norm(g[k].astype(np.float64)) for k in g.files if k.startswith("grad_f64_"))
    bad = []
    for name, p in net.named_parameters():
        got = p.grad.detach().cpu().numpy().astype(np.float64)
        r32, r64 = g["grad_" + name].astype(np.float64), g["grad_f64_" + name].astype(np.float64)
        n64 = np.linalg.norm(r64)
        if n64 < 1e-9 * gmax:  # a bias in front of a BatchNorm: zero true gradient
            assert np.linalg.norm(got) <= 1e-5 * gmax, name
            continue
        e32, e64, eref = np.linalg.norm(got - r32), np.linalg.norm(got - r64), np.linalg.norm(r32 - r64)
        print("[%s] grad %-22s vs ref fp32 %.2e  vs fp64 %.2e  (reference fp32 vs fp64 %.2e)" % (path, name, e32 / n64, e64 / n64, eref / n64))
        if e32 > 3e-3 * n64 or e64 > max(4 * eref, 3e-3 * n64):
            bad.append((name, e32 / n64, e64 / n64, eref / n64))
    assert not bad, bad
