"""Synchronised BatchNorm for the sampler's MLP (optional; SURVEY.md section 8e: "B = 256 single-process parity").

Data-parallel training keeps per-rank batch statistics by default (parallel.py: every replica behaves exactly like the
reference at its local batch size).  `convert_sync_batchnorm(net, process_group)` switches the nine BatchNorm layers of the
head to statistics over ALL ranks' rows, so that W ranks x B clouds train like one process with W B clouds:

  forward   per layer: GEMM (+ local sum / sum-of-squares partials) -> local (mean, biased variance) -> all_gather of
            (mean, variance, count) -> combined with the parallel-variance formula (no cancellation between a large mean and
            a small variance, which an all-reduce of raw sums would suffer on the FC head's nearly constant features) ->
            coefficients (scale, shift, mean, invstd) the next layer's operand load applies; running statistics updated with
            the global mean and the unbiased global variance;
  backward  per layer: the data / weight gradient kernels leave (sum dY, sum dY Z) partials of the BatchNorm below; dgamma /
            dbeta come from the LOCAL sums (the gradient all-reduce of the step averages them like every other parameter
            gradient), the dZ coefficients from the all-reduced sums and the global row count -- the split
            torch.nn.SyncBatchNorm makes.

It runs on the layer-by-layer entry points only (one collective per BatchNorm and direction: 18 per step -- the fused
single-node step, the FC chains and graph capture are per-rank-statistics paths and are bypassed), i.e. it is a parity mode, not
a fast path.  The collectives go through a small communicator object so that tests can stand two "ranks" up as threads of one
process on one GPU.
"""
import torch
import torch.distributed as dist

from ._lib import check, lib, ptr
from . import pointnet as P

DZ_PLAIN, DZ_BN, DZ_POOL = 0, 1, 2


class DistComm:
    """all_gather / all_reduce(SUM) of small fp64 / fp32 tensors over a torch.distributed process group."""

    def __init__(self, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("convert_sync_batchnorm needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)

    def all_gather(self, t):
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t.contiguous(), group=self.group)
        return torch.stack(out)

    def all_reduce_sum(self, t):
        t = t.contiguous().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def convert_sync_batchnorm(net, process_group=None, comm=None):
    """Switch the sampler's head to batch statistics over all ranks (training mode only; eval uses the running statistics as
    always).  Returns net.  comm: a communicator with all_gather / all_reduce_sum / world (default: DistComm(process_group))."""
    last = "bn_fc%d" % getattr(net, "num_fc_layers", 4)
    last_bn = getattr(net, last, None)
    if last_bn is not None and not isinstance(last_bn, torch.nn.SyncBatchNorm):
        # the classification sampler's BatchNorm on the head's OUTPUT is a torch module (pointnet_head applies it): torch's own
        # synchronised form takes its place (same parameter / buffer names)
        if comm is not None:
            raise NotImplementedError("convert_sync_batchnorm: the output BatchNorm (%s) needs a torch.distributed process group" % last)
        setattr(net, last, torch.nn.SyncBatchNorm.convert_sync_batchnorm(last_bn, process_group))
    net.__dict__["_sn_sync_bn"] = comm if comm is not None else DistComm(process_group)
    return net


def combine_stats(gathered, eps):
    """gathered (W, 3, C) fp64 rows (mean, biased variance, count) of every rank -> (mean, biased variance, total count) of the
    union of the ranks' rows (parallel-variance formula)."""
    n = gathered[:, 2, :1]                      # (W, 1): a rank's count is the same for every channel
    total = n.sum()
    mean = (gathered[:, 0] * n).sum(0) / total
    var = ((gathered[:, 1] + (gathered[:, 0] - mean) ** 2) * n).sum(0) / total
    return mean, var, total


def _global_coef(bn, coef_local, n_local, comm):
    """Local (scale, shift, mean, invstd) -> the same for the statistics of all ranks; running statistics updated."""
    eps = float(bn.eps)
    mean_l = coef_local[2].double()
    var_l = (coef_local[3].double() ** -2 - eps).clamp_min_(0.0)
    packed = torch.stack([mean_l, var_l, torch.full_like(mean_l, float(n_local))])
    mean, var, total = combine_stats(comm.all_gather(packed), eps)
    invstd = (var + eps).rsqrt()
    scale = bn.weight.detach().double() * invstd
    shift = bn.bias.detach().double() - mean * scale
    coef = torch.stack([scale, shift, mean, invstd]).float().contiguous()
    if bn.track_running_stats and bn.running_mean is not None:
        mom = P._momentum(bn)
        with torch.no_grad():
            unbiased = var * (total / (total - 1.0)) if float(total) > 1.0 else var
            bn.running_mean.mul_(1.0 - mom).add_(mean.float(), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(unbiased.float(), alpha=mom)
            bn.num_batches_tracked += 1
    return coef, int(total)


def _local_coef_from_partials(L, R, stats, nblk):
    """(scale, shift, mean, invstd) of the LOCAL rows from the GEMM's (sum, sum of squares) partials; no running update."""
    bn = L.bn
    coef = torch.empty(4, L.Co, device=L.W.device, dtype=torch.float32)
    check(lib.sn_bn_finalize(nblk, L.Co, R, ptr(stats), ptr(bn.weight), ptr(bn.bias), float(bn.eps), 0.0, None, None, None, ptr(coef),
                             P._st(L.W)), "sn_bn_finalize")
    return coef


def _local_coef_twopass(L, R, z):
    """The same from z itself in two passes (the FC head's rows: sum-of-squares partials lose digits there)."""
    bn = L.bn
    coef = torch.empty(4, L.Co, device=z.device, dtype=torch.float32)
    check(lib.sn_bn_batch_stats_twopass(R, L.Co, ptr(z), ptr(bn.weight), ptr(bn.bias), float(bn.eps), 0.0, None, None, None, ptr(coef),
                                        P._st(z)), "sn_bn_batch_stats_twopass")
    return coef


def forward_sync(net, x_bnc, comm):
    """Training forward with synchronised statistics: y (B, 3M) and what backward_sync needs."""
    convs, fcs = P._layers(net)
    B, N, _ = x_bnc.shape
    R = B * N
    saved = {"x": x_bnc, "B": B, "N": N, "zc": [], "cc": [], "zf": [], "cf": [], "rows_c": [], "rows_f": []}
    a_in, coef_prev = x_bnc.view(R, 3), None
    for L in convs:
        z, stats, nblk = P._linear_fwd(R, L, a_in, coef_prev, True)
        coef, total = _global_coef(L.bn, _local_coef_from_partials(L, R, stats, nblk), R, comm)
        saved["zc"].append(z), saved["cc"].append(coef), saved["rows_c"].append(total)
        a_in, coef_prev = z, coef
    C5 = convs[-1].Co
    pooled = P._empty((B, C5), x_bnc)
    argsel = P._empty((B, C5), x_bnc, torch.int32)
    zsel = P._empty((B, C5), x_bnc)
    check(lib.sn_pool_forward(B, N, C5, ptr(a_in), ptr(coef_prev), ptr(pooled), ptr(argsel), ptr(zsel), P._st(x_bnc)), "sn_pool_forward")
    saved.update(pooled=pooled, argsel=argsel, zsel=zsel)
    a_in, coef_prev = pooled, None
    for L in fcs[:-1]:
        z, _, _ = P._linear_fwd(B, L, a_in, coef_prev, False)
        if L.bn is None:
            coef, total = P._identity_coef(L.Co, z), 0
        else:
            coef, total = _global_coef(L.bn, _local_coef_twopass(L, B, z), B, comm)
        saved["zf"].append(z), saved["cf"].append(coef), saved["rows_f"].append(total)
        a_in, coef_prev = z, coef
    y, _, _ = P._linear_fwd(B, fcs[-1], a_in, coef_prev, False)
    return y, saved


def _bn_backward_sync(L, coef, stats_blocks, rows_global, comm, sink, bn_name, lin_name, grads):
    """BatchNorm backward of layer L from the (sum dY, sum dY Z) partials the kernels above it left: dgamma / dbeta from the local
    sums, the dZ coefficients (and the -- analytically zero -- bias gradient of the layer) from the sums of all ranks."""
    C = L.Co
    local = stats_blocks.reshape(-1, 2, C).sum(0).contiguous()
    dgamma, dbeta = P._out(sink, bn_name + ".weight", L.bn.weight), P._out(sink, bn_name + ".bias", L.bn.bias)
    scratch_k = torch.empty(3, C, device=local.device, dtype=torch.float32)
    check(lib.sn_bn_backward_coef(1, C, max(1, rows_global), ptr(local), ptr(coef), ptr(dgamma), ptr(dbeta), None, ptr(scratch_k),
                                  P._st(local)), "sn_bn_backward_coef")
    total = comm.all_reduce_sum(local)
    dbias = P._out(sink, lin_name + ".bias", L.b)
    kcoef = torch.empty(3, C, device=local.device, dtype=torch.float32)
    dg2, db2 = torch.empty(C, device=local.device), torch.empty(C, device=local.device)
    check(lib.sn_bn_backward_coef(1, C, rows_global, ptr(total), ptr(coef), ptr(dg2), ptr(db2), ptr(dbias), ptr(kcoef), P._st(local)),
          "sn_bn_backward_coef")
    grads[bn_name + ".weight"], grads[bn_name + ".bias"], grads[lin_name + ".bias"] = dgamma, dbeta, dbias
    return kcoef


def _dgrad_z(R, L, mode, dy, z, kcoef, zprev, coef_prev):
    """pointnet._dgrad with ZEROED statistics partials: which of the blocks a kernel variant fills depends on the route the shape
    takes inside the library (one per 64-row tile, or one per workgroup of the fused kernels); the sum over all of them must not
    see what it left unwritten."""
    dyprev = torch.empty(R, L.Ci, device=L.W.device, dtype=torch.float32)
    nblk = lib.sn_linear_stats_blocks(R)
    stats = torch.zeros(nblk, 2, L.Ci, device=L.W.device, dtype=torch.float32) if coef_prev is not None else None
    check(lib.sn_linear_dgrad(R, L.Ci, L.Co, mode, ptr(dy), ptr(z), ptr(kcoef), None, None, 1, ptr(L.W), ptr(zprev), ptr(coef_prev),
                              ptr(dyprev), ptr(stats), P._st(L.W)), "sn_linear_dgrad")
    return dyprev, stats


def _bwd_layer_z(R, L, mode, dy, z, kcoef, gsel, argsel, npts, zprev, coef_prev, sink, name):
    """dgrad + wgrad of one conv layer (sn_linear_backward), statistics partials zeroed (see _dgrad_z)."""
    dW = P._out(sink, name + ".weight", L.W)
    dyprev = torch.empty(R, L.Ci, device=L.W.device, dtype=torch.float32)
    stats = torch.zeros(lib.sn_linear_stats_blocks(R), 2, L.Ci, device=L.W.device, dtype=torch.float32)
    part = torch.empty(lib.sn_linear_wgrad_splits(R, L.Ci, L.Co, 0) * L.Co * L.Ci, device=L.W.device, dtype=torch.float32)
    check(lib.sn_linear_backward(R, L.Ci, L.Co, mode, ptr(dy), ptr(z), ptr(kcoef), ptr(gsel), ptr(argsel), npts, ptr(L.W), ptr(zprev),
                                 ptr(coef_prev), ptr(dyprev), ptr(stats), ptr(part), ptr(dW), P._st(L.W)), "sn_linear_backward")
    return dW, dyprev, stats


def backward_sync(net, saved, grad_y, comm, sink=None):
    """-> dict parameter name -> gradient (this rank's share: the step's gradient all-reduce averages them)."""
    convs, fcs = P._layers(net)
    B, N = saved["B"], saved["N"]
    R = B * N
    nf = len(fcs)
    zf, cf, zc, cc = saved["zf"], saved["cf"], saved["zc"], saved["cc"]
    grads = {}
    dy, kcoef = grad_y.contiguous(), None
    # ---- FC head: fc_last -> ... -> fc1 -> pooled features ----
    for j in range(nf - 1, -1, -1):
        L = fcs[j]
        last = j == nf - 1
        mode = DZ_PLAIN if (last or L.bn is None) else DZ_BN
        z = None if last else zf[j]
        kc = None if mode == DZ_PLAIN else kcoef
        aprev = zf[j - 1] if j > 0 else saved["pooled"]
        cprev = cf[j - 1] if j > 0 else None
        dW, db = P._wgrad(B, L, mode, dy, z, kc, None, None, 1, aprev, cprev, last or L.bn is None, sink, L.name)
        grads[L.name + ".weight"] = dW
        if db is not None:
            grads[L.name + ".bias"] = db
        dy, stats = _dgrad_z(B, L, mode, dy, z, kc, aprev, cprev)
        if j > 0 and fcs[j - 1].bn is not None:
            kcoef = _bn_backward_sync(fcs[j - 1], cf[j - 1], stats, saved["rows_f"][j - 1], comm, sink, fcs[j - 1].bn_name, fcs[j - 1].name,
                                      grads)
        else:
            kcoef = None
    # ---- max-pool + conv5's BatchNorm ----
    C5, L5 = convs[4].Co, convs[4]
    gsel = P._empty((B, C5), grad_y)
    pstats = P._empty((2 * C5,), grad_y)
    check(lib.sn_pool_backward(B, C5, ptr(dy), ptr(saved["pooled"]), ptr(saved["zsel"]), ptr(gsel), ptr(pstats), P._st(grad_y)),
          "sn_pool_backward")
    kcoef = _bn_backward_sync(L5, cc[4], pstats, saved["rows_c"][4], comm, sink, L5.bn_name, L5.name, grads)
    # ---- conv stack: conv5 -> conv2 (each leaves the sums of the BatchNorm below), conv1 ----
    dy = None
    for i in (4, 3, 2, 1):
        L = convs[i]
        mode = DZ_POOL if i == 4 else DZ_BN
        gs, ag = (gsel, saved["argsel"]) if i == 4 else (None, None)
        dW, dy, stats = _bwd_layer_z(R, L, mode, dy, zc[i], kcoef, gs, ag, N, zc[i - 1], cc[i - 1], sink, L.name)
        grads[L.name + ".weight"] = dW
        kcoef = _bn_backward_sync(convs[i - 1], cc[i - 1], stats, saved["rows_c"][i - 1], comm, sink, convs[i - 1].bn_name, convs[i - 1].name,
                                  grads)
    dW, _ = P._wgrad(R, convs[0], DZ_BN, dy, zc[0], kcoef, None, None, N, saved["x"].view(R, 3), None, False, sink, convs[0].name)
    grads[convs[0].name + ".weight"] = dW
    return grads


class SyncBNMLPFunction(torch.autograd.Function):
    """pointnet.PointNetMLPFunction with batch statistics over all ranks."""

    @staticmethod
    def forward(ctx, net, x_bnc, comm, *params):
        with torch.cuda.device(x_bnc.device):
            y, saved = forward_sync(net, x_bnc, comm)
        ctx.net, ctx.saved, ctx.comm = net, saved, comm
        return y

    @staticmethod
    def backward(ctx, grad_y):
        net = ctx.net
        sink, owner = P.sink_for_backward(net)
        with torch.cuda.device(grad_y.device):
            grads = backward_sync(net, ctx.saved, grad_y, ctx.comm, sink)
            if owner is not None:
                owner.commit(None if sink is not None else grads)
        return (None, None, None) + tuple(None if (owner is not None and n in owner) else grads[n] for n in P.param_order(net))
