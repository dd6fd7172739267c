"""GPU parity tests of the geometric kernels (pair scan, Chamfer, kNN, gathers, SoftProjection).

Every test calls the HIP path through the C ABI (samplenet_amd -> libsamplenet_hip.so) and compares with
the CPU oracle (oracle/) on the same seeded inputs and/or with the committed golden vectors that were
produced by running the reference modules (tests/golden/).  Bars: indices and Chamfer distances /
gradients bit-exact; soft-projection values within 1e-6 abs (1e-5 is the north-star bar).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def clouds(seed, b, n, m, dup=True):
    rng = np.random.default_rng(seed)
    x1 = rng.random((b, n, 3), dtype=np.float32) - 0.5
    x2 = rng.random((b, m, 3), dtype=np.float32) - 0.5
    if dup and n > 12 and m > 12:
        x2[:, 3] = x2[:, 9]      # exact duplicates: lowest index must win
        x2[:, 11] = x2[:, 9]
        x1[:, 2] = x1[:, 7]
        x1[0, 5] = x2[0, 4]      # zero distance
    return x1, x2


# ------------------------------------------------------------------------------------------ Chamfer
CH_SHAPES = [(2, 64, 1024), (3, 100, 37), (1, 1, 1), (2, 1, 70), (2, 513, 515), (4, 1024, 64), (2, 1024, 1024),
             (1, 2048, 64), (2, 64, 2048), (1, 2049, 70), (1, 70, 2049), (1, 3000, 2500), (32, 64, 1024)]


@pytest.mark.parametrize("shape", CH_SHAPES)
def test_chamfer_forward_backward_bit_exact(oracle, shape):
    from samplenet_amd import ops

    b, n, m = shape
    x1, x2 = clouds(n * 31 + m, b, n, m)
    o = oracle.chamfer_forward(x1, x2)
    t1, t2 = dev(x1).requires_grad_(True), dev(x2).requires_grad_(True)
    d1, d2, i1, i2 = ops.chamfer_distance(t1, t2, return_idx=True)
    assert np.array_equal(i1.cpu().numpy(), o[1]) and np.array_equal(i2.cpu().numpy(), o[3])
    assert np.array_equal(d1.detach().cpu().numpy(), o[0]) and np.array_equal(d2.detach().cpu().numpy(), o[2])
    rng = np.random.default_rng(1)
    g1 = rng.standard_normal((b, n)).astype(np.float32)
    g2 = rng.standard_normal((b, m)).astype(np.float32)
    og1, og2 = oracle.chamfer_backward(x1, x2, g1, o[1], g2, o[3])
    gx1, gx2 = torch.autograd.grad([d1, d2], [t1, t2], [dev(g1), dev(g2)])
    assert np.array_equal(gx1.cpu().numpy(), og1)
    assert np.array_equal(gx2.cpu().numpy(), og2)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_chamfer_module_matches_reference_golden(golden, tag):
    from samplenet_amd import ChamferDistance

    g = golden("chamfer_reference.npz")
    t1 = dev(g[f"{tag}_xyz1"]).requires_grad_(True)
    t2 = dev(g[f"{tag}_xyz2"]).requires_grad_(True)
    d1, d2 = ChamferDistance()(t1, t2)
    assert np.array_equal(d1.detach().cpu().numpy(), g[f"{tag}_dist1"])
    assert np.array_equal(d2.detach().cpu().numpy(), g[f"{tag}_dist2"])
    gx1, gx2 = torch.autograd.grad([d1, d2], [t1, t2], [dev(g[f"{tag}_gdist1"]), dev(g[f"{tag}_gdist2"])])
    assert np.array_equal(gx1.cpu().numpy(), g[f"{tag}_gxyz1"])
    assert np.array_equal(gx2.cpu().numpy(), g[f"{tag}_gxyz2"])


def test_chamfer_empty_and_errors():
    from samplenet_amd import ops
    from samplenet_amd._lib import SampleNetHipError

    d1, d2 = ops.chamfer_distance(torch.zeros(0, 5, 3).cuda(), torch.zeros(0, 7, 3).cuda())
    assert d1.shape == (0, 5) and d2.shape == (0, 7)
    with pytest.raises(SampleNetHipError):
        ops.chamfer_distance(torch.zeros(2, 0, 3).cuda(), torch.zeros(2, 7, 3).cuda())
    with pytest.raises(RuntimeError):
        ops.chamfer_distance(torch.zeros(2, 4, 3), torch.zeros(2, 7, 3))  # CPU tensors: no fallback


def test_chamfer_full_size_properties():
    """BASELINE config 2 size (B=32, 1024 <-> 64) and a saturating batch: size-independent properties."""
    from samplenet_amd import ops

    for B in (32, 1024):
        g = torch.Generator(device="cuda").manual_seed(B)
        ref = torch.rand(B, 1024, 3, device="cuda", generator=g) - 0.5
        smp = torch.rand(B, 64, 3, device="cuda", generator=g) - 0.5
        d1, d2, i1, i2 = ops.chamfer_distance(smp, ref, return_idx=True)
        # (1) the reported distance is exactly the distance to the reported index
        nn1 = torch.gather(ref, 1, i1.long().unsqueeze(2).expand(-1, -1, 3))
        diff = nn1 - smp
        re = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        assert torch.equal(re, d1)
        # (2) swapping the arguments swaps the outputs bit-for-bit
        e2, e1, j2, j1 = ops.chamfer_distance(ref, smp, return_idx=True)
        assert torch.equal(e1, d1) and torch.equal(e2, d2) and torch.equal(j1, i1) and torch.equal(j2, i2)
        # (3) minimality against the dense matrix of a few clouds
        D = ((smp[:4, :, None, :] - ref[:4, None, :, :]) ** 2)
        D = (D[..., 0] + D[..., 1]) + D[..., 2]
        assert torch.equal(D.min(2)[0], d1[:4]) and torch.equal(D.min(1)[0], d2[:4])


# ------------------------------------------------------------------------------------------ kNN
@pytest.mark.parametrize("cfg", [(3, 1024, 64, 8), (2, 1024, 64, 7), (2, 2048, 64, 16), (2, 500, 40, 1), (1, 6, 9, 3),
                                 (2, 64, 64, 64), (1, 100, 5, 33), (2, 3000, 50, 8), (1, 5000, 20, 16), (2, 1024, 1024, 8),
                                 (1, 200, 3, 10)])
@pytest.mark.parametrize("layout", ["bcn", "bnc"])
def test_knn_matches_oracle(oracle, cfg, layout):
    from samplenet_amd import ops

    b, n, m, k = cfg
    P, Q = clouds(n + m + k, b, n, m)
    if n > 300:
        P[:, 100:140] = P[:, 100:101]  # 40 identical points: deep ties at the k-th boundary
        Q[:, 0] = P[:, 100]
    od, oi = oracle.knn(k, P, Q)
    if layout == "bcn":
        idx, d2 = ops.knn(k, dev(P.transpose(0, 2, 1)), dev(Q.transpose(0, 2, 1)), ops.BCN, ops.BCN)
    else:
        idx, d2 = ops.knn(k, dev(P), dev(Q), ops.BNC, ops.BNC)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(d2.cpu().numpy(), od)


@pytest.mark.parametrize("n", [700, 714, 1024])
def test_knn_all_points_identical(oracle, n):
    """Every point passes the threshold filter: the candidate list overflows and the compaction falls back to its merging form.
    n = 714 ends that form with a merge followed by a round of 10 candidates (26 entries: the rank-by-counting path behind a
    fallback), n = 700 with 140 entries (the merge path), n = 1024 is the variant without bound checks."""
    from samplenet_amd import ops

    P = np.full((2, n, 3), 0.25, np.float32)
    Q = np.random.default_rng(0).random((2, 9, 3), dtype=np.float32)
    od, oi = oracle.knn(16, P, Q)
    idx, d2 = ops.knn(16, dev(P), dev(Q), ops.BNC, ops.BNC)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d2.cpu().numpy(), od)
    assert np.array_equal(oi[0, 0], np.arange(16))


@pytest.mark.parametrize("n,k", [(1024, 8), (1024, 16), (1000, 8), (2048, 16)])
@pytest.mark.parametrize("cluster", [70, 130, 150, 191, 192, 193, 260])
def test_knn_cluster_of_coincident_points(oracle, n, k, cluster):
    """`cluster` copies of one point, spread over every lane group of the scan, sit nearest to all queries: exactly that many
    candidates pass the threshold filter -- sizes either side of one candidate per lane (64), of the point where a round of
    the branch-free compaction runs past the list's end (128..192) and of the list itself (192: the checked form takes over).
    Ties resolve to the lowest indices; the fused scan's other products must agree with the oracle too."""
    from samplenet_amd import ops

    rng = np.random.default_rng(cluster * 7 + k)
    b, m = 3, 40
    P = (rng.random((b, n, 3), dtype=np.float32) - 0.5) * 0.2 + 2.0   # far away
    c = np.array([0.1, -0.2, 0.05], np.float32)
    for bi in range(b):
        lanes = np.arange(64) + 64 * rng.integers(0, n // 64, 64)  # a copy in every lane (point i sits in lane i % 64)
        rest = rng.permutation(np.setdiff1d(np.arange(n), lanes))[:cluster - 64]
        P[bi, np.concatenate([lanes, rest])] = c
        assert int((P[bi] == c).all(1).sum()) == cluster
    Q = (c + 0.01 * rng.standard_normal((b, m, 3))).astype(np.float32)
    od, oi = oracle.knn(k, P, Q)
    idx, d2 = ops.knn(k, dev(P), dev(Q), ops.BNC, ops.BNC)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d2.cpu().numpy(), od)
    Pc, Qc = np.ascontiguousarray(P.transpose(0, 2, 1)), np.ascontiguousarray(Q.transpose(0, 2, 1))
    oproj, _, _ = oracle.softproj_forward(Pc, Qc, oi, 1.0)
    proj, idx2, dq, iq, dp, ip = ops.SoftProjectFunction.apply(dev(Pc), dev(Qc), torch.tensor(1.0, device="cuda"), 1e-2, k, True)
    assert np.array_equal(idx2.cpu().numpy(), oi)
    np.testing.assert_allclose(proj.cpu().numpy(), oproj, rtol=0, atol=1e-6)
    ocd = oracle.chamfer_forward(Q, P)
    assert np.array_equal(dq.cpu().numpy(), ocd[0]) and np.array_equal(iq.cpu().numpy(), ocd[1])
    assert np.array_equal(dp.cpu().numpy(), ocd[2]) and np.array_equal(ip.cpu().numpy(), ocd[3])


def test_knn_in_tree_definition_when_distinct(oracle):
    """Membership and order equal the in-tree TF definition (matrix + selection sort) on distinct distances."""
    from samplenet_amd import ops

    P, Q = clouds(77, 2, 1024, 64, dup=False)
    _, ti = oracle.knn_point_tf(8, P, Q)
    idx, _ = ops.knn(8, dev(P), dev(Q), ops.BNC, ops.BNC)
    assert np.array_equal(idx.cpu().numpy(), ti)


def test_knn_compat_shim_interface(oracle):
    from samplenet_amd.compat.knn_cuda import KNN

    P, Q = clouds(5, 2, 300, 20)
    d, i = KNN(4, transpose_mode=False)(dev(P.transpose(0, 2, 1)), dev(Q.transpose(0, 2, 1)))
    od, oi = oracle.knn(4, P, Q)
    assert d.shape == (2, 4, 20) and i.dtype == torch.int64
    assert np.array_equal(i.permute(0, 2, 1).cpu().numpy(), oi)
    np.testing.assert_allclose(d.permute(0, 2, 1).cpu().numpy(), np.sqrt(od), rtol=1e-6)
    d, i = KNN(4, transpose_mode=True)(dev(P), dev(Q))
    assert np.array_equal(i.cpu().numpy(), oi)


def test_cd_plugin_surface_with_caller_allocated_outputs(oracle):
    """samplenet_amd.compat.cd = the reference's pybind module `cd` (chamfer_distance.cpp:180-185) over the C ABI, driven the way
    registration/src/chamfer_distance/chamfer_distance.py:14-66 drives it: the CALLER allocates every output (zeros, as the
    reference does) and the functions write them in place.  Bit-exact against the oracle, forward and backward; a side
    stream is honoured; CPU tensors raise."""
    from samplenet_amd.compat import cd

    rng = np.random.default_rng(4)
    for (B, n, m) in [(32, 64, 1024), (3, 100, 37)]:
        a, b = rng.random((B, n, 3), dtype=np.float32) - 0.5, rng.random((B, m, 3), dtype=np.float32) - 0.5
        xyz1, xyz2 = dev(a), dev(b)
        # chamfer_distance.py:21-34
        dist1, dist2 = torch.zeros(B, n).cuda(), torch.zeros(B, m).cuda()
        idx1, idx2 = torch.zeros(B, n, dtype=torch.int).cuda(), torch.zeros(B, m, dtype=torch.int).cuda()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            assert cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2) is None
        torch.cuda.current_stream().wait_stream(side)
        od1, oi1, od2, oi2 = oracle.chamfer_forward(a, b)
        assert np.array_equal(dist1.cpu().numpy(), od1) and np.array_equal(idx1.cpu().numpy(), oi1)
        assert np.array_equal(dist2.cpu().numpy(), od2) and np.array_equal(idx2.cpu().numpy(), oi2)
        # chamfer_distance.py:44-59
        g1, g2 = rng.standard_normal((B, n)).astype(np.float32), rng.standard_normal((B, m)).astype(np.float32)
        gradxyz1, gradxyz2 = torch.zeros(xyz1.size()).cuda(), torch.zeros(xyz2.size()).cuda()
        assert cd.backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, dev(g1), dev(g2), idx1, idx2) is None
        og1, og2 = oracle.chamfer_backward(a, b, g1, oi1, g2, oi2)
        assert np.array_equal(gradxyz1.cpu().numpy(), og1) and np.array_equal(gradxyz2.cpu().numpy(), og2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        cd.forward(xyz1.cpu(), xyz2.cpu(), dist1.cpu(), dist2.cpu(), idx1.cpu(), idx2.cpu())
    with pytest.raises(RuntimeError):
        cd.forward_cuda(xyz1.cpu(), xyz2, dist1, dist2, idx1, idx2)
    with pytest.raises(RuntimeError):
        cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1.long(), idx2)


def test_reference_call_pattern_over_the_compat_packages(oracle):
    """INTEGRATION level 1: the reference's SoftProjection.project (registration/src/soft_projection.py:11-14, 75-99, 138-152),
    restated here call for call in plain torch -- knn_cuda.KNN(k, transpose_mode=False)(ref, query), the permute / int cast,
    pointnet2_utils.grouping_operation, softmax over the group -- with the two third-party names resolved through
    samplenet_amd.compat.install().  (The reference file itself cannot be imported on the GPU box: /root/reference is not
    there.)  Same projection and same gradients as the fused kernel path and the oracle."""
    import importlib

    from samplenet_amd import SoftProjection, compat

    compat.install()
    KNN = importlib.import_module("knn_cuda").KNN
    grouping_operation = importlib.import_module("pointnet2.utils.pointnet2_utils").grouping_operation
    rng = np.random.default_rng(12)
    B, N, M, K, sigma = 4, 1024, 64, 8, 0.09
    P, Q = rng.random((B, 3, N), dtype=np.float32) - 0.5, rng.random((B, 3, M), dtype=np.float32) - 0.5
    pc, qc = dev(P), dev(Q).requires_grad_(True)
    # soft_projection.py:11-14 + 75-90
    _, idx = KNN(K, transpose_mode=False)(pc.contiguous(), qc.detach().contiguous())
    idx = idx.permute(0, 2, 1).type(torch.int32)
    grouped = grouping_operation(pc, idx.contiguous())                                   # (B,3,M,K)
    # :92-95, 138-152
    deltas = grouped - qc.unsqueeze(-1).expand_as(grouped)
    dist = torch.sum(deltas ** 2, dim=1, keepdim=True) / sigma
    weights = torch.softmax(-dist, dim=3).repeat(1, 3, 1, 1)
    proj = torch.sum(grouped * weights, dim=3)
    go = dev(rng.standard_normal((B, 3, M)).astype(np.float32))
    (gq,) = torch.autograd.grad(proj, qc, go)
    _, oidx = oracle.knn(K, P.transpose(0, 2, 1), Q.transpose(0, 2, 1))
    assert np.array_equal(idx.cpu().numpy(), oidx)
    oproj, _, _ = oracle.softproj_forward(P, Q, oidx, sigma)
    np.testing.assert_allclose(proj.detach().cpu().numpy(), oproj, rtol=0, atol=1e-6)
    sp = SoftProjection(K, initial_temperature=0.3, min_sigma=1e-4).cuda()
    q2 = dev(Q).requires_grad_(True)
    proj2 = sp(pc, q2)
    (gq2,) = torch.autograd.grad(proj2, q2, go)
    np.testing.assert_allclose(proj2.detach().cpu().numpy(), proj.detach().cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(gq2.cpu().numpy(), gq.cpu().numpy(), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------ gathers
@pytest.mark.parametrize("shape", [(3, 50, 6, 11, 4), (2, 1024, 3, 64, 8), (2, 300, 5, 130, 3), (1, 70, 1, 1, 1)])
def test_group_point_and_grad(oracle, shape):
    """The gradient adds the hits of a row in ascending entry order, the reference's CPU loop: bit-identical to the oracle (and
    from run to run -- the reference's GPU kernel and round 1's used unordered float atomics)."""
    from samplenet_amd import ops

    b, n, c, m, ns = shape
    rng = np.random.default_rng(2)
    pts = rng.random((b, n, c), dtype=np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    idx[:, : max(1, m // 2)] = idx[:, :1]  # many entries on the same few rows
    t = dev(pts).requires_grad_(True)
    out = ops.group_point(t, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.group_point(pts, idx))
    go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    (g,) = torch.autograd.grad(out, t, dev(go))
    assert np.array_equal(g.cpu().numpy(), oracle.group_point_grad(pts.shape, idx, go))


def test_large_grouping_gradients_take_the_atomic_route():
    """PointNet++-sized grouping (n = 16384 rows, 16384 x 32 indices per cloud): the ordered scan would cost (n / 64) x ne index
    loads per cloud, so beyond 2^26 of those the gradient is scattered with float atomics into a zeroed destination, like the
    reference's own kernels (tf_grouping_g.cu:60-78) -- same sums, summation order free: compared with torch.index_add_ in fp64."""
    from samplenet_amd import ops

    b, n, c, m, ns = 2, 16384, 4, 16384, 32
    assert (n // 64) * m * ns > 1 << 26
    g = torch.Generator(device="cuda").manual_seed(8)
    idx = torch.randint(0, n, (b, m, ns), device="cuda", generator=g, dtype=torch.int32)
    go = torch.randn(b, m, ns, c, device="cuda", generator=g)
    pts = torch.zeros(b, n, c, device="cuda", requires_grad=True)
    (gp,) = torch.autograd.grad(ops.group_point(pts, idx), pts, go)
    want = torch.zeros(b, n, c, device="cuda", dtype=torch.float64)
    for i in range(b):
        want[i].index_add_(0, idx[i].reshape(-1).long(), go[i].reshape(-1, c).double())
    assert float((gp.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    feat = torch.zeros(b, c, n, device="cuda", requires_grad=True)
    go2 = go.permute(0, 3, 1, 2).contiguous()  # (b, c, m, ns)
    (gf,) = torch.autograd.grad(ops.grouping_operation(feat, idx), feat, go2)
    assert float((gf.double() - want.permute(0, 2, 1)).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("shape", [(2, 5, 64, 9, 8), (2, 3, 1024, 64, 8), (1, 7, 333, 200, 2)])
def test_grouping_operation_and_grad(oracle, shape):
    from samplenet_amd.compat.pointnet2.utils.pointnet2_utils import grouping_operation

    b, c, n, m, ns = shape
    rng = np.random.default_rng(3)
    feat = rng.random((b, c, n), dtype=np.float32)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    idx[:, : max(1, m // 2)] = idx[:, :1]
    t = dev(feat).requires_grad_(True)
    out = grouping_operation(t, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.grouping_operation(feat, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    (g,) = torch.autograd.grad(out, t, dev(go))
    assert np.array_equal(g.cpu().numpy(), oracle.grouping_operation_grad(feat.shape, idx, go))


@pytest.mark.parametrize("cfg", [(4, 1024, 64, 8, 1.0), (2, 1024, 64, 7, 0.3), (2, 2048, 64, 16, 0.05), (1, 33, 7, 16, 1.0),
                                 (3, 200, 24, 1, 0.5),
                                 # the progressive sampler's output sizes at N = 1024
                                 (2, 1024, 256, 8, 1.0), (2, 1024, 128, 8, 0.3), (3, 1024, 32, 8, 1.0), (32, 1024, 256, 7, 0.1),
                                 # C4's sampler shape: 2048 -> 64, K = 16 at a full batch of 50
                                 (50, 2048, 64, 16, 1.0)])
def test_soft_project_fused_vs_oracle(oracle, cfg):
    from samplenet_amd import ops

    b, n, m, k, T = cfg
    Pn, Qn = clouds(n * 3 + k, b, n, m)
    # queries near the surface make near-ties likely (SURVEY 8d set ii)
    Qn = (Pn[:, :m] + 0.02 * np.random.default_rng(k).standard_normal((b, m, 3))).astype(np.float32)
    P = np.ascontiguousarray(Pn.transpose(0, 2, 1))
    Q = np.ascontiguousarray(Qn.transpose(0, 2, 1))
    min_sigma = 1e-2
    sigma = max(T * T, min_sigma)
    _, oi = oracle.knn(k, Pn, Qn)
    oproj, _, _ = oracle.softproj_forward(P, Q, oi, sigma)
    tP, tQ = dev(P).requires_grad_(True), dev(Q).requires_grad_(True)
    tT = torch.tensor(T, device="cuda", requires_grad=True)
    proj, idx, dq, iq, dp, ip = ops.SoftProjectFunction.apply(tP, tQ, tT, min_sigma, k, True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_allclose(proj.detach().cpu().numpy(), oproj, rtol=0, atol=1e-6)
    od = oracle.chamfer_forward(Qn, Pn)
    assert np.array_equal(dq.cpu().numpy(), od[0]) and np.array_equal(iq.cpu().numpy(), od[1])
    assert np.array_equal(dp.cpu().numpy(), od[2]) and np.array_equal(ip.cpu().numpy(), od[3])
    gp = np.random.default_rng(9).standard_normal(oproj.shape).astype(np.float32)
    ogq, ogp, ogs = oracle.softproj_backward(P, Q, oi, sigma, gp, want_grad_P=True)
    gP, gQ, gT = torch.autograd.grad(proj, [tP, tQ, tT], dev(gp), retain_graph=True)
    np.testing.assert_allclose(gQ.cpu().numpy(), ogq, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gP.cpu().numpy(), ogp, rtol=1e-4, atol=1e-5)
    # the gradient towards the point cloud is summed per point in the reference loop's order (sn_soft_project_backward_ordered:
    # no float atomics): the same bits on every run, and as close to the oracle as the gradient towards the query (the
    # per-neighbour terms differ from the oracle's by the exponential's last bit, not by their order of addition)
    gP2, gQ2, _ = torch.autograd.grad(proj, [tP, tQ, tT], dev(gp))
    assert torch.equal(gP, gP2) and torch.equal(gQ, gQ2)
    assert np.abs(gP.cpu().numpy() - ogp).max() <= 4 * max(np.abs(gQ.cpu().numpy() - ogq).max(), 1e-7)
    ogT = ogs * 2 * T if T * T > min_sigma else 0.0
    np.testing.assert_allclose(float(gT), ogT, rtol=1e-4, atol=1e-5)


def test_propagation_gradients_are_deterministic(oracle):
    """SoftProjection.project_and_propagate (soft_projection.py:101-120): gradients towards the point cloud AND the features
    come out of ordered per-point sums (sn_soft_weights_backward_ordered / sn_weighted_gather_backward_ordered) -- bit-equal
    from run to run, and equal to torch's own deterministic index_add composition of the same contributions to 1e-6."""
    from samplenet_amd.soft_projection import SoftProjection

    b, n, m, k, c = 4, 512, 96, 8, 24
    Pn, Qn = clouds(77, b, n, m)
    Qn = (Pn[:, :m] + 0.02 * np.random.default_rng(1).standard_normal((b, m, 3))).astype(np.float32)
    P = dev(np.ascontiguousarray(Pn.transpose(0, 2, 1))).requires_grad_(True)
    Q = dev(np.ascontiguousarray(Qn.transpose(0, 2, 1))).requires_grad_(True)
    Fe = torch.randn(b, c, n, device="cuda", requires_grad=True)
    sp = SoftProjection(k, initial_temperature=0.5).cuda()
    runs = []
    for _ in range(3):
        pp, pf = sp(P, Q, Fe, action="project_and_propagate")
        g = torch.autograd.grad([pp, pf], [P, Q, Fe], [torch.ones_like(pp) * 0.3, torch.cos(pf.detach())])
        runs.append(g)
    for g in runs[1:]:
        for u, v in zip(runs[0], g):
            assert torch.equal(u, v)
    # same contributions through torch: weights w (b,m,k) and neighbours idx -> features gradient by index_add in fp64
    _, oi = oracle.knn(k, Pn, Qn)
    idx = torch.from_numpy(oi).cuda().long()
    with torch.no_grad():
        grouped = torch.gather(P.detach().unsqueeze(2).expand(b, 3, m, n), 3, idx.unsqueeze(1).expand(b, 3, m, k))
        d = ((grouped - Q.detach().unsqueeze(-1)) ** 2).sum(1) / sp.sigma()
        w = torch.softmax(-d, dim=2).double()                                     # (b, m, k)
        gout = torch.cos(pf.detach()).double()                                    # (b, c, m)
        contrib = gout.unsqueeze(-1) * w.unsqueeze(1)                             # (b, c, m, k)
        ref = torch.zeros(b, c, n, device="cuda", dtype=torch.float64)
        ref.scatter_add_(2, idx.reshape(b, 1, m * k).expand(b, c, m * k), contrib.reshape(b, c, m * k))
    assert float((runs[0][2].double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


def test_fused_scan_when_the_batch_fills_the_chip(oracle):
    """B >= 512: one workgroup per cloud, four waves with 16 queries each (pairscan_dispatch) -- every product of the scan
    against the oracle on all 512 clouds."""
    from samplenet_amd import ops

    b, n, m, k = 512, 1024, 64, 8
    Pn, _ = clouds(91, b, n, m)
    Qn = (Pn[:, :m] + 0.02 * np.random.default_rng(3).standard_normal((b, m, 3))).astype(np.float32)
    P = np.ascontiguousarray(Pn.transpose(0, 2, 1))
    Q = np.ascontiguousarray(Qn.transpose(0, 2, 1))
    _, oi = oracle.knn(k, Pn, Qn)
    oproj, _, _ = oracle.softproj_forward(P, Q, oi, 1.0)
    proj, idx, dq, iq, dp, ip = ops.SoftProjectFunction.apply(dev(P), dev(Q), torch.tensor(1.0, device="cuda"), 1e-2, k, True)
    assert np.array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_allclose(proj.cpu().numpy(), oproj, rtol=0, atol=1e-6)
    od = oracle.chamfer_forward(Qn, Pn)
    assert np.array_equal(dq.cpu().numpy(), od[0]) and np.array_equal(iq.cpu().numpy(), od[1])
    assert np.array_equal(dp.cpu().numpy(), od[2]) and np.array_equal(ip.cpu().numpy(), od[3])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_soft_projection_module_matches_reference_golden(golden, tag):
    """All three actions + gradients against outputs of the reference module (tests/golden/make_golden.py)."""
    from samplenet_amd import SoftProjection

    g = golden("softproj_reference.npz")
    K, T = int(g[f"{tag}_K"]), float(g[f"{tag}_T"])
    sp = SoftProjection(K, initial_temperature=T, is_temperature_trainable=True, min_sigma=1e-2).cuda()
    P = dev(g[f"{tag}_P"]).requires_grad_(True)
    Q = dev(g[f"{tag}_Q"]).requires_grad_(True)
    Fe = dev(g[f"{tag}_F"]).requires_grad_(True)
    proj = sp(P, Q, action="project")
    np.testing.assert_allclose(proj.detach().cpu().numpy(), g[f"{tag}_proj"], rtol=0, atol=1e-6)
    gP, gQ, gT = torch.autograd.grad(proj, [P, Q, sp._temperature], dev(g[f"{tag}_gproj"]))
    np.testing.assert_allclose(gQ.cpu().numpy(), g[f"{tag}_gQ"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gP.cpu().numpy(), g[f"{tag}_gP"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gT.cpu().numpy(), g[f"{tag}_gT"], rtol=1e-4, atol=1e-5)
    prop = sp(P, Q, Fe, action="propagate")
    np.testing.assert_allclose(prop.detach().cpu().numpy(), g[f"{tag}_prop"], rtol=1e-6, atol=2e-6)
    hP, hQ, hF, hT = torch.autograd.grad(prop, [P, Q, Fe, sp._temperature], dev(g[f"{tag}_gprop"]))
    np.testing.assert_allclose(hQ.cpu().numpy(), g[f"{tag}_hQ"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(hP.cpu().numpy(), g[f"{tag}_hP"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(hF.cpu().numpy(), g[f"{tag}_hF"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(hT.cpu().numpy(), g[f"{tag}_hT"], rtol=1e-4, atol=2e-5)
    proj2, prop2 = sp(P, Q, Fe, action="project_and_propagate")
    np.testing.assert_allclose(proj2.detach().cpu().numpy(), g[f"{tag}_proj2"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(prop2.detach().cpu().numpy(), g[f"{tag}_prop2"], rtol=1e-6, atol=2e-6)
    with pytest.raises(ValueError):
        sp(P, Q, action="nope")


def test_soft_projection_known_answers(golden):
    """registration/src/soft_projection.py:158-284 and classification/soft_projection.py:86-161 through the module."""
    from samplenet_amd import SoftProjection

    g = golden("known_answer_registration.npz")
    P = dev(g["point_cloud"].T[None], torch.float32)
    Q = dev(g["query_cloud"].T[None], torch.float32)
    Fe = dev(g["point_features"].T[None], torch.float32)
    sp = SoftProjection(3, initial_temperature=1.0).cuda()
    prop = sp.propagate(P, Fe, Q).cpu().detach().numpy().squeeze()
    assert np.mean(np.sum((prop.T - g["expected_features_nn_3"]) ** 2, axis=1)) < 1e-6
    sd = sp.state_dict()
    sd["_temperature"] = torch.tensor(0.1, dtype=torch.float32)
    sp.load_state_dict(sd)
    proj = sp.project(Q, P).cpu().detach().numpy().squeeze()
    assert np.mean(np.sum((proj.T - g["expected_nn_cloud"]) ** 2, axis=1)) < 1e-6

    c = golden("known_answer_classification.npz")
    pc = np.stack([c["point_cloud"], c["point_cloud"] * 3]).astype(np.float32)
    qc = np.stack([c["query_cloud"], c["query_cloud"] * 3]).astype(np.float32)
    exp_soft = np.stack([c["expected_cloud_soft"], c["expected_cloud_soft"] * 3])
    sp = SoftProjection(3, initial_temperature=0.01, min_sigma=1e-4).cuda()
    out = sp(dev(pc.transpose(0, 2, 1)), dev(qc.transpose(0, 2, 1)))
    assert np.abs(out.detach().cpu().numpy().transpose(0, 2, 1) - exp_soft).max() < 2e-3


def test_soft_project_full_size_properties():
    """B=32 (config 2) and B=512: projected points lie in the convex hull of their neighbours; T -> 0 gives the
    nearest neighbour (hard projection, classification/soft_projection.py:73-76); idempotence on dataset points."""
    from samplenet_amd import ops

    for B in (32, 512):
        g = torch.Generator(device="cuda").manual_seed(B)
        P = (torch.rand(B, 3, 1024, device="cuda", generator=g) - 0.5)
        Q = (torch.rand(B, 3, 64, device="cuda", generator=g) - 0.5)
        T = torch.tensor(1.0, device="cuda")
        proj, idx = ops.SoftProjectFunction.apply(P, Q, T, 1e-2, 8, False)
        nb = torch.gather(P.unsqueeze(2).expand(-1, -1, 64, -1), 3, idx.long().unsqueeze(1).expand(-1, 3, -1, -1))
        assert torch.all(proj <= nb.max(3)[0] + 1e-6) and torch.all(proj >= nb.min(3)[0] - 1e-6)
        tiny = torch.tensor(1e-10, device="cuda")  # sigma = 1e-20: only exact distance ties could share weight
        hard, idx1 = ops.SoftProjectFunction.apply(P, Q, tiny, 1e-30, 8, False)
        assert torch.allclose(hard, nb[..., 0], atol=1e-6)
        # queries that ARE dataset points project onto themselves as T -> 0
        Q2 = P[:, :, 5:69].contiguous()
        self_proj, idx2 = ops.SoftProjectFunction.apply(P, Q2, tiny, 1e-30, 8, False)
        assert torch.equal(idx2[:, :, 0], torch.arange(5, 69, device="cuda", dtype=torch.int32).expand(B, -1))
        assert torch.allclose(self_proj, Q2, atol=1e-6)


# ------------------------------------------------------------------------------------------ fused simplification loss
@pytest.mark.parametrize("shape", [(32, 64, 1024), (3, 12, 96), (1, 1, 7), (5, 300, 200)])
def test_fused_simplification_loss_matches_composition(oracle, shape):
    """ops.SimplificationLossFunction == the reference's op-by-op composition (samplenet.py:171-181) on the same
    Chamfer products: value within 1e-6 relative of the fp64 evaluation on the oracle's distances, gradients equal to
    autograd through the explicit-gradient Chamfer backward within fp32 rounding."""
    from samplenet_amd import ops

    b, m, n = shape
    smp, ref = clouds(m * 13 + n, b, m, n)
    w = 1.0 + 0.25 * 64
    ts, tr = dev(smp).requires_grad_(True), dev(ref).requires_grad_(True)
    _, _, d1, i1, d2, i2 = ops.chamfer_forward_impl(ts.detach(), tr.detach())
    loss = ops.SimplificationLossFunction.apply(ts, tr, d1, i1, d2, i2, w)
    od1, _, od2, _ = oracle.chamfer_forward(smp, ref)
    oloss = od1.mean(dtype=np.float64) + od1.max(1).mean(dtype=np.float64) + w * od2.mean(dtype=np.float64)
    assert abs(float(loss.detach()) - oloss) <= 1e-6 * max(1.0, abs(oloss))
    gs, gr = torch.autograd.grad(loss, [ts, tr], torch.tensor(0.7, device="cuda"))
    e1, e2 = ops.chamfer_distance(ts, tr)
    ref_loss = e1.mean() + e1.max(1)[0].mean() + w * e2.mean()
    hs, hr = torch.autograd.grad(ref_loss, [ts, tr], torch.tensor(0.7, device="cuda"))
    np.testing.assert_allclose(gs.cpu().numpy(), hs.cpu().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gr.cpu().numpy(), hr.cpu().numpy(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("shape", [(32, 64, 1024), (3, 12, 96), (2, 1, 7), (5, 300, 200)])
def test_fused_simplification_loss_channel_major_sample(shape):
    """samp_layout = BCN (the FC head's (B,3,M) output, used by SampleNet.get_simplification_loss on a scan hit):
    same loss, and the gradient is bit-for-bit the transposed gradient of the (B,M,3) call."""
    from samplenet_amd import ops

    b, m, n = shape
    smp, ref = clouds(m * 7 + n, b, m, n)
    ts, tr = dev(smp).requires_grad_(True), dev(ref)
    tsT = dev(np.ascontiguousarray(smp.transpose(0, 2, 1))).requires_grad_(True)
    _, _, d1, i1, d2, i2 = ops.chamfer_forward_impl(ts.detach(), tr)
    la = ops.SimplificationLossFunction.apply(ts, tr, d1, i1, d2, i2, 1.5)
    lb = ops.SimplificationLossFunction.apply(tsT, tr, d1, i1, d2, i2, 1.5, ops.BCN)
    assert float(la.detach()) == float(lb.detach())
    (ga,) = torch.autograd.grad(la, [ts], torch.tensor(0.3, device="cuda"))
    (gb,) = torch.autograd.grad(lb, [tsT], torch.tensor(0.3, device="cuda"))
    assert gb.shape == tsT.shape
    assert torch.equal(ga.permute(0, 2, 1), gb)


# ------------------------------------------------------------------------------------------ device-side inference matching (row f2)
@pytest.mark.parametrize("B,N,k,dup", [(4, 1024, 64, 0.4), (3, 1500, 100, 0.7), (2, 100, 32, 0.9), (5, 4100, 64, 0.2),
                                       (2, 64, 64, 0.0)])
@pytest.mark.parametrize("layout", ["bnc", "bcn"])
def test_nn_matching_device_equals_numpy(oracle, B, N, k, dup, layout):
    """ops.nn_matching (sn_nn_matching: first-occurrence unique + farthest-point completion in one kernel per cloud)
    returns exactly the points of the reference's numpy routine (sputils.py:7-41, restated in samplenet_amd/sputils.py
    and oracle.nn_matching): float64 distances, first-maximum argmax -- for many duplicates, none, N not a multiple of
    the workgroup size, and both cloud layouts; complete_fps=False is the plain gather."""
    from samplenet_amd import ops, sputils

    rng = np.random.default_rng(B * 1000 + N + k)
    pc = (rng.random((B, N, 3), dtype=np.float32) - 0.5).astype(np.float32)
    idx = rng.integers(0, N, size=(B, k))
    ndup = int(dup * k)
    if ndup:
        idx[:, -ndup:] = idx[:, :ndup]  # repeat earlier picks
        idx = np.stack([rng.permutation(row) for row in idx])
    ref = sputils.nn_matching(pc, idx, k, complete_fps=True)
    ora = oracle.nn_matching(pc, idx.astype(np.int64), k, True)
    assert np.array_equal(ref, ora)
    x = dev(pc) if layout == "bnc" else dev(np.ascontiguousarray(pc.transpose(0, 2, 1)))
    lay = ops.BNC if layout == "bnc" else ops.BCN
    got = ops.nn_matching(x, torch.from_numpy(idx).cuda(), k, True, lay).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == (B, k, 3)
    assert np.array_equal(got.astype(np.float64), ref)
    plain = ops.nn_matching(x, torch.from_numpy(idx).cuda(), k, False, lay).cpu().numpy()
    assert np.array_equal(plain.astype(np.float64), sputils.nn_matching(pc, idx, k, complete_fps=False))


def test_nn_matching_device_equals_reference_golden(golden):
    """the golden vectors produced by the reference's own sputils.nn_matching (tests/golden/make_golden.py)."""
    from samplenet_amd import ops

    g = golden("nn_matching_reference.npz")
    k = g["idx"].shape[1]
    x, idx = dev(g["pc"].astype(np.float32)), torch.from_numpy(g["idx"]).cuda()
    assert np.array_equal(ops.nn_matching(x, idx, k, True).cpu().numpy().astype(np.float64), g["out_fps"])
    assert np.array_equal(ops.nn_matching(x, idx, k, False).cpu().numpy().astype(np.float64), g["out_nofps"])


def test_hard_projection_is_nearest_input_point(oracle):
    """SoftProjection.project(hard=True) (row f4; TF semantics classification/soft_projection.py:73-76: one_hot(argmax) of the
    softmax weights): every query lands on its nearest input point -- bit-exact with the oracle's 1-NN gather."""
    from samplenet_amd import SoftProjection

    rng = np.random.default_rng(3)
    pc = (rng.random((4, 700, 3), dtype=np.float32) - 0.5).astype(np.float32)
    q = (rng.random((4, 50, 3), dtype=np.float32) - 0.5).astype(np.float32)
    sp = SoftProjection(8, initial_temperature=0.3).cuda()
    P = dev(np.ascontiguousarray(pc.transpose(0, 2, 1)))
    Q = dev(np.ascontiguousarray(q.transpose(0, 2, 1)))
    out = sp.project(P, Q, hard=True)
    _, idx = oracle.knn(1, pc, q)
    want = np.take_along_axis(pc, idx[:, :, :1].astype(np.int64).repeat(3, axis=2), axis=1)  # (B,M,3)
    assert np.array_equal(out.permute(0, 2, 1).cpu().numpy(), want)


@pytest.mark.parametrize("B,N,M,sizes", [(4, 1024, 256, [32, 64, 128, 256]), (2, 300, 50, [1, 7, 50]), (32, 1024, 64, [8, 16, 32, 64]),
                                          (1, 2048, 1024, [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]), (3, 777, 21, [5, 6, 20, 21]),
                                          (300, 1024, 16, [8, 16])])  # (the last: the one-thread-per-point kernel of large batches)
def test_prefix_point_minima_match_chamfer_per_prefix(oracle, B, N, M, sizes):
    """sn_prefix_point_minima (progressive sampler): the per-point Chamfer products of every nested prefix from ONE
    pass == the oracle's Chamfer scan of that prefix, bit for bit (distances and first-minimum indices, duplicated points
    included); the full-size slice also equals the HIP pair scan's."""
    from samplenet_amd import ops

    rng = np.random.default_rng(N + M)
    P = (rng.random((B, N, 3), dtype=np.float32) - 0.5)
    Q = (P[:, rng.permutation(N)[:M]] + 0.02 * rng.standard_normal((B, M, 3))).astype(np.float32)
    Q[:, M // 2] = Q[:, 0]  # duplicated query: the lower index must win in every prefix that holds both
    d, i = ops.prefix_point_minima(dev(P), dev(Q), sizes)
    for j, s in enumerate(sizes):
        _, _, od2, oi2 = oracle.chamfer_forward(np.ascontiguousarray(Q[:, :s]), P)
        assert np.array_equal(d[j].cpu().numpy(), od2), s
        assert np.array_equal(i[j].cpu().numpy(), oi2), s
    _, _, _, _, hd2, hi2 = ops.chamfer_forward_impl(dev(Q), dev(P))
    assert torch.equal(d[-1], hd2) and torch.equal(i[-1], hi2)


@pytest.mark.parametrize("B,N,M,K,kind", [(512, 1024, 64, 8, "random"), (512, 1024, 64, 7, "near"), (600, 1000, 64, 16, "clusters"),
                                          (512, 2048, 64, 16, "random"), (512, 1024, 64, 8, "identical"), (700, 900, 33, 3, "near")])
def test_large_batch_scan_matches_the_oracle(oracle, B, N, M, K, kind):
    """The pair scan's LARGE-BATCH launch shape (B >= 512: one workgroup of four waves per cloud, every wave walks >= 16 queries --
    another launch configuration than the split clouds of B = 32) on random clouds, near-surface queries (near-ties), coincident
    clusters and all-identical clouds (ties resolve to the lowest index): kNN indices / distances and both Chamfer directions of a
    sample of clouds equal the oracle bit for bit, and two runs agree on every cloud."""
    import numpy as np
    import torch

    from samplenet_amd import ops

    g = torch.Generator(device="cuda").manual_seed(B + N + K)
    P = torch.rand(B, N, 3, device="cuda", generator=g) - 0.5
    if kind == "near":
        perm = torch.randperm(N, device="cuda", generator=g)[:M]
        Q = P[:, perm] + 0.02 * torch.randn(B, M, 3, device="cuda", generator=g)
    elif kind == "clusters":
        P = P[:, : N // 8].repeat(1, 8, 1).contiguous()  # every point eight times
        Q = torch.rand(B, M, 3, device="cuda", generator=g) - 0.5
    elif kind == "identical":
        P = P[:, :1].expand(B, N, 3).contiguous()
        Q = torch.rand(B, M, 3, device="cuda", generator=g) - 0.5
    else:
        Q = torch.rand(B, M, 3, device="cuda", generator=g) - 0.5
    P, Q = P.contiguous(), Q.contiguous()

    def run():
        idx, dist = ops.knn(K, P, Q, ops.BNC, ops.BNC)
        d1, d2, i1, i2 = ops.ChamferDistanceFunction.apply(Q, P)
        torch.cuda.synchronize()
        return idx, dist, d1, d2, i1, i2

    a, b = run(), run()
    for i, (u, w) in enumerate(zip(a, b)):
        assert torch.equal(u, w), (kind, i)
    sel = [0, B // 2, B - 1]
    od, oi = oracle.knn(K, P[sel].cpu().numpy(), Q[sel].cpu().numpy())
    assert np.array_equal(a[0][sel].cpu().numpy(), oi) and np.array_equal(a[1][sel].cpu().numpy(), od)
    c1, ci1, c2, ci2 = oracle.chamfer_forward(Q[sel].cpu().numpy(), P[sel].cpu().numpy())
    # (ChamferDistanceFunction returns dist1, dist2, idx1, idx2)
    assert np.array_equal(a[2][sel].cpu().numpy(), c1) and np.array_equal(a[4][sel].cpu().numpy(), ci1)
    assert np.array_equal(a[3][sel].cpu().numpy(), c2) and np.array_equal(a[5][sel].cpu().numpy(), ci2)


@pytest.mark.parametrize("shape", [(3, 64, 3), (32, 256, 3), (5, 100, 1), (2, 1000, 7)])
def test_prefix_pack_and_its_gradient_equal_the_slices(shape):
    """ops.prefix_pack (sn_prefix_pack / sn_prefix_scatter_sum: the progressive sampler's nested prefixes, classification/
    train_samplenet_progressive.py:157-234, as contiguous tensors from one launch): bit-equal to the slices; the gradient equals
    autograd through the slices (same ascending accumulation order -> bit-equal), also when a prefix receives no gradient; int32
    payloads (the scan's indices) travel unchanged."""
    from samplenet_amd import ops

    B, M, C = shape
    sizes = sorted({1, M // 4, M // 2, M - 1, M} - {0})
    g = torch.Generator(device="cuda").manual_seed(5)
    t = torch.randn(B, M, C, device="cuda", generator=g, requires_grad=True)
    outs = ops.prefix_pack(t, sizes)
    for o, s in zip(outs, sizes):
        assert o.is_contiguous() and torch.equal(o, t[:, :s, :])
    w = [torch.randn(B, s, C, device="cuda", generator=g) for s in sizes]
    for skip in (None, 1):
        terms = [(o * wi).sum() for j, (o, wi) in enumerate(zip(outs, w)) if j != skip]
        refs = [(t[:, :s, :] * wi).sum() for j, (s, wi) in enumerate(zip(sizes, w)) if j != skip]
        ga, = torch.autograd.grad(sum(terms), t, retain_graph=True)
        acc = torch.zeros_like(t)  # ascending prefix order, zero-padded: what the kernel sums
        for j, (s, wi) in enumerate(zip(sizes, w)):
            if j != skip:
                acc[:, :s, :] += wi
        assert torch.equal(ga, acc)
        gb, = torch.autograd.grad(sum(refs), t, retain_graph=True)
        assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-6)
    idx = torch.randint(0, 1 << 30, (B, M), device="cuda", dtype=torch.int32, generator=g)
    for o, s in zip(ops.prefix_pack(idx, sizes), sizes):
        assert o.dtype == torch.int32 and torch.equal(o, idx[:, :s])
    with pytest.raises(RuntimeError):
        ops.prefix_pack(t.detach(), [M + 1])


@pytest.mark.parametrize("cfg", [(4, 200, 64, [8, 16, 32]), (32, 1024, 256, [32, 64, 128]), (3, 1500, 96, [5, 96]), (2, 60, 40, [1, 7, 39])])
def test_prefix_simplification_loss_is_bit_identical_to_the_separate_terms(cfg):
    """ops.PrefixSimplificationLossFunction (the progressive sampler's prefix terms, classification/train_samplenet_progressive.py:
    204-216, as one node) against SimplificationLossFunction term by term on contiguous copies of the prefixes, added ascending:
    the same loss and the same gradient on the simplified cloud, bit for bit."""
    from samplenet_amd import ops

    B, N, M, sizes = cfg
    g = torch.Generator(device="cuda").manual_seed(B + N)
    ref = torch.rand(B, N, 3, device="cuda", generator=g) - 0.5
    samp = (torch.rand(B, M, 3, device="cuda", generator=g) - 0.5).requires_grad_(True)
    all_sizes = sizes if sizes[-1] == M else sizes + [M]
    _, _, dq, iq, _, _ = ops.chamfer_forward_impl(samp.detach(), ref)
    d2, i2 = ops.prefix_point_minima(ref, samp, all_sizes)
    w = [1.0 + 0.02 * s for s in sizes]
    fused = ops.PrefixSimplificationLossFunction.apply(samp, ref, dq, iq, d2, i2, sizes, w)
    tot = None
    for j, s in enumerate(sizes):
        term = ops.SimplificationLossFunction.apply(samp[:, :s, :].contiguous(), ref, dq[:, :s].contiguous(), iq[:, :s].contiguous(),
                                                    d2[j], i2[j], w[j])
        tot = term if tot is None else tot + term
    assert torch.equal(fused, tot)
    ga, = torch.autograd.grad(fused * 0.37, samp)
    gb, = torch.autograd.grad(tot * 0.37, samp)
    assert torch.allclose(ga, gb, rtol=1e-5, atol=1e-9)
    # bit for bit: every term's own gradient, added largest prefix first (the order autograd adds them in)
    acc = torch.zeros_like(samp)
    for j in reversed(range(len(sizes))):
        sl = samp.detach()[:, :sizes[j], :].contiguous().requires_grad_(True)
        term = ops.SimplificationLossFunction.apply(sl, ref, dq[:, :sizes[j]].contiguous(), iq[:, :sizes[j]].contiguous(), d2[j], i2[j], w[j])
        gj, = torch.autograd.grad(term * 0.37, sl)
        acc[:, :sizes[j], :] += gj
    assert torch.equal(ga, acc)
    assert float(ga[:, sizes[-1]:].abs().sum()) == 0.0


@pytest.mark.parametrize("T,ms", [(1.0, 1e-2), (0.05, 1e-2), (0.1, 1e-2), (-0.7, 1e-2), (0.3, 0.0), (0.0, 0.0)])
def test_sigma_op_equals_torch(T, ms):
    """SoftProjection.sigma() (registration/src/soft_projection.py:97-99: max(T^2, min_sigma), what get_projection_loss returns) as
    ops.SigmaFunction -- one launch each way -- against the torch expression: the same value and the same gradient, bit for bit,
    incl. the floor (no gradient), the exact tie (torch.max splits the gradient evenly) and a negative temperature."""
    from samplenet_amd import SoftProjection

    sp = SoftProjection(8, initial_temperature=T, is_temperature_trainable=True, min_sigma=ms).cuda()
    s = sp.sigma()
    t = sp._temperature.detach().clone().requires_grad_(True)
    ref = torch.max(t ** 2, torch.tensor(ms, device="cuda", dtype=torch.float32))
    assert s.shape == ref.shape and torch.equal(s.detach(), ref.detach())
    (g,) = torch.autograd.grad(0.37 * s, sp._temperature)
    (gr,) = torch.autograd.grad(0.37 * ref, t)
    assert torch.equal(g, gr), (float(g), float(gr))
