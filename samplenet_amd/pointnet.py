"""PointNet feature extractor + FC head of SampleNet on the hand-written MFMA kernels
(samplenet_amd/csrc/pointnet_mlp.hip, pointnet_mlp_backward.hip, fc_chain.hip), as ONE autograd node.

Replaces the torch.nn call chain of registration/src/samplenet.py:90-104
    5 x relu(bn(conv1d_k1(.)))  ->  max over points  ->  3 x relu(bn(linear(.)))  ->  linear
while the parameters stay ordinary nn.Conv1d / nn.BatchNorm1d / nn.Linear members of the module
(state_dict compatibility).  Host side: buffer allocation and launch sequencing only.
"""
import torch

from ._lib import check, lib, ptr, stream_of

DZ_PLAIN, DZ_BN, DZ_POOL = 0, 1, 2
SYNC_STRIDE = 32  # words between the words of a chain launch's sync state (sn_common.h: SN_FC_SYNC_STRIDE): [i, 0] is word i


def _st(t):
    return stream_of(t)


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


class _Layer:
    """Plain record of one GEMM layer's tensors (name / bn_name: the module attribute names = state_dict prefixes)."""

    __slots__ = ("name", "bn_name", "W", "b", "bn", "Ci", "Co")

    def __init__(self, name, lin, bn_name, bn):
        self.name, self.bn_name = name, bn_name
        self.W, self.b, self.bn = lin.weight, lin.bias, bn
        self.Co, self.Ci = lin.weight.shape[0], lin.weight.shape[1]


def _layers(net):
    """conv1..conv5 (each with its BatchNorm) and fc1..fcK of the sampler.  Hidden FC layers carry a BatchNorm in the
    registration / classification samplers (samplenet.py:52-59) and none in the reconstruction sampler (samplers.py:33-38);
    the last FC layer is always returned without one -- a BatchNorm behind it (classification/models/samplenet_model.py:
    100-108; no activation) is a record of its own (out_bn): forward_impl / backward_impl run it on sn_layer_forward_bn_out /
    sn_bn_output_backward, its weight and bias close the parameter list.
    The records are kept on the module (this runs several times per step and nn.Module attribute lookups are slow) and
    rebuilt when a layer's weight Parameter is no longer the one recorded (a layer was replaced)."""
    d = net.__dict__
    mods = d["_modules"]
    cached = d.get("_sn_layer_records")
    out_bn = _plain_out_bn(net, mods)
    if cached is not None:
        convs, fcs = cached[0], cached[1]
        ok = (cached[4][1] if cached[4] is not None else None) is out_bn
        for L in convs + fcs:  # weight, bias and the BatchNorm MODULE are still the recorded ones (a replaced bias / a layer swapped
            m = mods[L.name]   # for torch.nn.SyncBatchNorm or a frozen copy would otherwise keep feeding stale pointers)
            if m._parameters["weight"] is not L.W or m._parameters["bias"] is not L.b or \
                    (L.bn is not None and mods.get(L.bn_name) is not L.bn):
                ok = False
                break
        if ok:
            return convs, fcs
    convs = [_Layer("conv%d" % i, getattr(net, "conv%d" % i), "bn%d" % i, getattr(net, "bn%d" % i)) for i in range(1, 6)]
    nfc = getattr(net, "num_fc_layers", 4)
    fcs = [_Layer("fc%d" % i, getattr(net, "fc%d" % i), "bn_fc%d" % i, getattr(net, "bn_fc%d" % i, None) if i < nfc else None)
           for i in range(1, nfc + 1)]
    names = []
    for L in convs + fcs:
        names += [L.name + ".weight", L.name + ".bias"]
    for L in convs + fcs:
        if L.bn is not None:
            names += [L.bn_name + ".weight", L.bn_name + ".bias"]
    params = []
    for L in convs + fcs:
        params += [L.W, L.b]
    for L in convs + fcs:
        if L.bn is not None:
            params += [L.bn.weight, L.bn.bias]
    ob = None
    if out_bn is not None:  # the BatchNorm behind the last FC layer: differentiated by the node as well (out_bn below)
        ob = ("bn_fc%d" % nfc, out_bn)
        names += [ob[0] + ".weight", ob[0] + ".bias"]
        params += [out_bn.weight, out_bn.bias]
    d["_sn_layer_records"] = (convs, fcs, tuple(names), tuple(params), ob)
    return convs, fcs


def _plain_out_bn(net, mods):
    """The BatchNorm module behind the LAST FC layer when the HIP path applies it itself: a plain torch.nn.BatchNorm1d of the
    classification sampler (classification/models/samplenet_model.py:100-108).  None: no such layer, or one that torch applies on
    the head's output (torch.nn.SyncBatchNorm after syncbn.convert_sync_batchnorm: statistics over all ranks)."""
    m = mods.get("bn_fc%d" % net.__dict__.get("num_fc_layers", 4))
    if m is None or type(m) is not torch.nn.BatchNorm1d or not m.affine or net.__dict__.get("_sn_sync_bn") is not None:
        return None
    return m


def out_bn(net):
    """(module attribute name, BatchNorm1d) of the output BatchNorm the head's node applies and differentiates, or None."""
    _layers(net)
    return net.__dict__["_sn_layer_records"][4]


def param_order(net):
    """Names of the parameters the MLP node differentiates, in the order they are passed to / returned from it."""
    _layers(net)
    return net.__dict__["_sn_layer_records"][2]


def param_list(net):
    """The parameters param_order names, in that order."""
    _layers(net)
    return net.__dict__["_sn_layer_records"][3]


_IDENT = {}


def _identity_coef(C, like):
    """BatchNorm coefficient block (scale, shift, mean, invstd) = (1, 0, 0, 1): a ReLU layer without BatchNorm, expressed in
    the operand / backward modes of the GEMM kernels (with "fixed statistics" rows < 0 its backward is dZ = relu' * dY)."""
    key = (C, like.device)
    t = _IDENT.get(key)
    if t is None:
        t = torch.zeros(4, C, device=like.device, dtype=torch.float32)
        t[0].fill_(1.0)
        t[3].fill_(1.0)
        _IDENT[key] = t
    return t


def _linear_fwd(R, L, a_in, coef_prev, want_stats, fc_rows=False):
    z = _empty((R, L.Co), a_in)
    if fc_rows and not want_stats and R > 32:  # a layer of the FC head above 32 clouds (no statistics asked of the GEMM)
        check(lib.sn_linear_forward_rows(R, L.Ci, L.Co, ptr(a_in), ptr(coef_prev), ptr(L.W), ptr(L.b), ptr(z), _st(a_in)),
              "sn_linear_forward_rows")
        return z, None, 0
    nblk = lib.sn_linear_stats_blocks(R)
    stats = _empty((nblk, 2, L.Co), a_in) if want_stats else None
    check(lib.sn_linear_forward(R, L.Ci, L.Co, ptr(a_in), ptr(coef_prev), ptr(L.W), ptr(L.b), ptr(z), ptr(stats), _st(a_in)),
          "sn_linear_forward")
    return z, stats, nblk


def _momentum(bn):
    """torch.nn.BatchNorm semantics: momentum=None means a cumulative moving average, factor 1 / (num_batches_tracked + 1)
    for this step (the kernels increment the counter themselves)."""
    if bn.momentum is not None:
        return float(bn.momentum)
    if not bn.track_running_stats or bn.num_batches_tracked is None:
        return 0.0
    return 1.0 / (float(bn.num_batches_tracked.item()) + 1.0)  # (host read: not capturable; momentum=None is rare)


def _layer_fwd_bn(R, L, a_in, coef_prev):
    """Training forward of a layer with BatchNorm: pre-BN output z and coef (scale, shift, mean, invstd); running
    statistics updated in place.  One launch when R <= 32 (finalisation fused), GEMM + bn_finalize otherwise."""
    bn = L.bn
    z = _empty((R, L.Co), a_in)
    coef = _empty((4, L.Co), a_in)
    stats = _empty((lib.sn_linear_stats_blocks(R), 2, L.Co), a_in)
    mom = _momentum(bn)
    upd = bn.track_running_stats
    check(lib.sn_layer_forward_bn(R, L.Ci, L.Co, ptr(a_in), ptr(coef_prev), ptr(L.W), ptr(L.b), ptr(z), ptr(stats),
                                  ptr(bn.weight), ptr(bn.bias), float(bn.eps), float(mom),
                                  ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None,
                                  ptr(bn.num_batches_tracked) if upd else None, ptr(coef), _st(a_in)), "sn_layer_forward_bn")
    return z, coef


def _layer_fwd_bn_pool(R, npts, L, a_in, coef_prev, pooled, argsel, zsel):
    """_layer_fwd_bn of the last conv layer + the max-pool over the points (sn_conv_forward_bn_pool)."""
    bn = L.bn
    z = _empty((R, L.Co), a_in)
    coef = _empty((4, L.Co), a_in)
    nblk = lib.sn_linear_stats_blocks(R)
    stats = _empty((nblk, 2, L.Co), a_in)
    pool_val = _empty((nblk, 2, L.Co), a_in)
    pool_idx = _empty((nblk, 2, L.Co), a_in, torch.int32)
    mom = _momentum(bn)
    upd = bn.track_running_stats
    check(lib.sn_conv_forward_bn_pool(R, L.Ci, L.Co, npts, ptr(a_in), ptr(coef_prev), ptr(L.W), ptr(L.b), ptr(z), ptr(stats),
                                      ptr(bn.weight), ptr(bn.bias), float(bn.eps), float(mom),
                                      ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None,
                                      ptr(bn.num_batches_tracked) if upd else None, ptr(coef), ptr(pool_val), ptr(pool_idx),
                                      ptr(pooled), ptr(argsel), ptr(zsel), _st(a_in)), "sn_conv_forward_bn_pool")
    return z, coef


def _bn_coef(L, R, stats, nblk, training):
    bn = L.bn
    C = L.Co
    coef = _empty((4, C), L.W)
    if training or not bn.track_running_stats:
        mom = _momentum(bn)
        upd = training and bn.track_running_stats
        check(lib.sn_bn_finalize(nblk, C, R, ptr(stats), ptr(bn.weight), ptr(bn.bias), float(bn.eps), float(mom),
                                 ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None,
                                 ptr(bn.num_batches_tracked) if upd else None, ptr(coef), _st(L.W)), "sn_bn_finalize")
    else:
        check(lib.sn_bn_eval_coef(C, ptr(bn.weight), ptr(bn.bias), float(bn.eps), ptr(bn.running_mean), ptr(bn.running_var),
                                  ptr(coef), _st(L.W)), "sn_bn_eval_coef")
    return coef


def _conv_stack_fx(net, convs, x_bnc, B, N, saved, pooled, argsel, zsel, defer_pool=False, rec=None):
    """Training forward of the conv stack + max-pool as one call (sn_conv_stack_forward_bn: batch statistics as fixed-point
    sums, every layer finalises the BatchNorm of its input -- no reduction launch between layers).  Fills saved["zc"] /
    saved["cc"]; returns False when the shapes are not supported (the per-layer path runs instead).
    defer_pool: stop after the last GEMM -- the last BatchNorm and the pool pick run as the first stage of the FC chain
    (_fc_chain_fwd with saved["pool_tail"])."""
    import ctypes

    n = len(convs)
    chans = (ctypes.c_int * (n + 1))(convs[0].Ci, *[L.Co for L in convs])
    if any(L.bn.momentum is None or not L.bn.track_running_stats for L in convs):
        return False
    if not lib.sn_conv_stack_forward_supported(B, N, n, chans):
        return False
    if defer_pool and max(chans) > 128:  # (the FC chain's pool stage reads the 128-channel accumulator layout)
        return False
    R = B * N
    acc = getattr(net, "_fx_acc", None)
    nacc = lib.sn_conv_stack_acc_elems(n)
    if acc is None or acc.device != x_bnc.device or acc.numel() != nacc:
        acc = torch.zeros(nacc, device=x_bnc.device, dtype=torch.int64)  # persistent: every call leaves it zero
        net._fx_acc = acc
    zs = [_empty((R, L.Co), x_bnc) for L in convs]
    if Z1_FREE and lib.sn_conv_stack_z1_free_supported(B, N, n, chans):
        # the xyz layer's activation tensor is never written: conv2's forward and backward rebuild it from the cloud (same
        # expression, bit for bit); the per-layer backward materialises it on demand (_z1)
        zs[0] = None
    cs = [_empty((4, L.Co), x_bnc) for L in convs]
    Cn = convs[-1].Co
    nblk = lib.sn_linear_stats_blocks(R)
    # (defer_pool: the last layer leaves (B, 2, Cn) 64-bit keys here instead of the block partials)
    pool_val = _empty((max(nblk, 2 * B) if defer_pool else nblk, 2, Cn), x_bnc)
    pool_idx = _empty((nblk, 2, Cn), x_bnc, torch.int32)
    VP = ctypes.c_void_p * n

    def arr(ts):
        return VP(*[ptr(t) for t in ts])

    eps = (ctypes.c_float * n)(*[float(L.bn.eps) for L in convs])
    mom = (ctypes.c_float * n)(*[float(L.bn.momentum) for L in convs])
    args = [B, N, n, chans, ptr(x_bnc), arr([L.W for L in convs]), arr([L.b for L in convs]),
            arr([L.bn.weight for L in convs]), arr([L.bn.bias for L in convs]),
            arr([L.bn.running_mean for L in convs]), arr([L.bn.running_var for L in convs]),
            arr([L.bn.num_batches_tracked for L in convs]), eps, mom, arr(zs), arr(cs), ptr(acc),
            ptr(pool_val), ptr(pool_idx), *([None] * 3 if defer_pool else [ptr(pooled), ptr(argsel), ptr(zsel)]), _st(x_bnc)]
    try:
        check(lib.sn_conv_stack_forward_bn(*args), "sn_conv_stack_forward_bn")
    except Exception:
        acc.zero_()  # a launch failed half way: do not leave partial sums behind
        raise
    if rec is not None:  # (forward plan, see _ForwardPlan: argument 4 = the cloud, the last one = the stream)
        rec.append((lib.sn_conv_stack_forward_bn, "sn_conv_stack_forward_bn", args, 4, acc, (pool_val, pool_idx)))
    saved["zc"], saved["cc"] = zs, cs
    if defer_pool:
        saved["pool_tail"] = (acc, pool_val, pool_idx, convs[-1], n, N, argsel, zsel)
    return True


def _fc_chain_shape(hidden, B):
    """(C0, H, n) when sn_fc_chain_forward can run these hidden FC layers on B rows, else None."""
    n = len(hidden)
    if n < 2 or any(L.bn is None or L.bn.momentum is None or not L.bn.track_running_stats for L in hidden):
        return None
    H, C0 = hidden[0].Co, hidden[0].Ci
    if any(L.Co != H for L in hidden) or any(L.Ci != H for L in hidden[1:]):
        return None
    if not lib.sn_fc_chain_forward_supported(B, C0, H, n):
        return None
    return C0, H, n


def _fc_chain_fwd(net, hidden, pooled, B, saved, rec=None, out=None):
    """The FC head's BatchNorm + ReLU layers as one launch (sn_fc_chain_forward; with saved["pool_tail"] from a deferred
    conv stack: sn_fc_chain_forward_pool, which also finishes the last conv BatchNorm and the max-pool).  Fills saved["zf"] /
    saved["cf"]; returns False when the shape is not supported (the per-layer launches run instead).
    out = (last FC layer, out_bn record): the classification sampler's output layer + BatchNorm as the chain's last stage
    (sn_fc_chain_forward_pool_out) where the shape allows -- saved["y_out"] then holds the head's output and saved["z_out"] /
    saved["c_out"] what backward_impl needs; otherwise the caller runs _last_layer."""
    import ctypes

    shape = _fc_chain_shape(hidden, B)
    if shape is None:
        return False
    C0, H, n = shape
    sync = getattr(net, "_fc_sync", None)
    if sync is None or sync.device != pooled.device:
        sync = torch.zeros(16, SYNC_STRIDE, device=pooled.device, dtype=torch.int32)  # persistent: epoch + monotonic arrival counters
        net._fc_sync = sync
    xbuf = _empty((2 * 32 * H,), pooled)
    zs = [_empty((B, H), pooled) for _ in hidden]
    cs = [_empty((4, H), pooled) for _ in hidden]
    VP = ctypes.c_void_p * n

    def arr(ts):
        return VP(*[ptr(t) for t in ts])

    eps = (ctypes.c_float * n)(*[float(L.bn.eps) for L in hidden])
    mom = (ctypes.c_float * n)(*[float(L.bn.momentum) for L in hidden])
    layer_args = (arr([L.W for L in hidden]), arr([L.b for L in hidden]),
                  arr([L.bn.weight for L in hidden]), arr([L.bn.bias for L in hidden]),
                  arr([L.bn.running_mean for L in hidden]), arr([L.bn.running_var for L in hidden]),
                  arr([L.bn.num_batches_tracked for L in hidden]), eps, mom, arr(zs), arr(cs), ptr(xbuf), ptr(sync), _st(pooled))
    tail = saved.pop("pool_tail", None)
    if tail is not None:
        acc, pool_val, pool_idx, L5, nconv, N, argsel, zsel = tail
        bn5 = L5.bn
        args = [B, N, nconv, ptr(acc), ptr(pool_val), ptr(pool_idx), ptr(bn5.weight), ptr(bn5.bias),
                ptr(bn5.running_mean), ptr(bn5.running_var), ptr(bn5.num_batches_tracked),
                float(bn5.eps), float(bn5.momentum), ptr(saved["cc"][-1]), ptr(pooled), ptr(argsel),
                ptr(zsel), H, n, *layer_args]
        fn, name, keep = lib.sn_fc_chain_forward_pool, "sn_fc_chain_forward_pool", ()
        if out is not None:
            Lo, (_, obn) = out
            if (Lo.Ci == H and obn.momentum is not None and lib.sn_fc_chain_forward_pool_out_supported(B, N, C0, H, n, Lo.Co)):
                upd = obn.track_running_stats
                zo, co, yo = _empty((B, Lo.Co), pooled), _empty((4, Lo.Co), pooled), _empty((B, Lo.Co), pooled)
                args = args[:-1] + [Lo.Co, ptr(Lo.W), ptr(Lo.b), ptr(obn.weight), ptr(obn.bias),
                                    ptr(obn.running_mean) if upd else None, ptr(obn.running_var) if upd else None,
                                    ptr(obn.num_batches_tracked) if upd else None, float(obn.eps), float(obn.momentum),
                                    ptr(zo), ptr(co), ptr(yo), args[-1]]
                fn, name, keep = lib.sn_fc_chain_forward_pool_out, "sn_fc_chain_forward_pool_out", (zo, co, yo)
                saved["z_out"], saved["c_out"], saved["out_fixed"], saved["y_out"] = zo, co, False, yo
        try:
            check(fn(*args), name)
        except Exception:
            acc.zero_()
            raise
        if rec is not None:
            rec.append((fn, name, args, None, acc, keep))
    else:
        check(lib.sn_fc_chain_forward(B, C0, H, n, ptr(pooled), *layer_args), "sn_fc_chain_forward")
    saved["zf"], saved["cf"] = zs, cs
    saved["fc_chain"] = xbuf  # (scratch of the asynchronous launch)
    return True


_CHAIN_ERRORS = {1: "forward seam 0", 2: "forward seam 1", 3: "forward seam 2", 64: "epoch hand-over of the backward chain"}


def chain_error_words(net):
    """(forward, backward) error words of the module's FC chain launches as Python ints -- a HOST READ (synchronises the
    stream): 0 = healthy.  See check_chain_errors."""
    out = []
    for name in ("_fc_sync", "_fc_sync_b"):
        t = getattr(net, name, None)
        out.append(int(t[15, 0].item()) if t is not None else 0)
    return tuple(out)


def check_chain_errors(net, raise_error=True):
    """The FC chain kernels (sn_fc_chain_forward / _backward: 8 and 16 workgroups that hand activations to each other inside
    ONE launch) need their workgroups resident together; a device kept full by other work (another process, a collective's
    persistent kernels on every CU) can starve one of them, the others' polls then give up instead of hanging the GPU.  The
    kernels make that visible on the device -- NaN in place of their outputs, hence a NaN loss and NaN gradients, and a NaN loss
    value from the fused step's tail -- and here on the host: a non-zero error word raises SampleNetHipError after the launch
    state was re-armed (counters zeroed; the next step runs normally).  Costs a device synchronisation: call it where the loss
    is read back anyway (engine.SamplerTrainStep.check(), bench.py after the timed loop), not per step."""
    from ._lib import SampleNetHipError

    words = chain_error_words(net)
    if not any(words):
        return False
    for name in ("_fc_sync", "_fc_sync_b"):
        t = getattr(net, name, None)
        if t is not None:
            limit = int(t[13, 0].item())
            t.zero_()  # epoch + monotonic arrival counters are out of step after a timeout: start over
            t[13, 0] = limit
    if raise_error:
        what = ["%s chain: %s" % (d, _CHAIN_ERRORS.get(w, "hand-off %d" % w)) for d, w in zip(("forward", "backward"), words) if w]
        raise SampleNetHipError("FC chain launch timed out waiting for a co-resident workgroup (%s); its outputs were poisoned "
                                "with NaN -- this step's loss / gradients are invalid.  The device was too busy to hold the "
                                "chain's workgroups together (8 / 16 workgroups x 137 KB LDS); the launch state was reset."
                                % "; ".join(what))
    return True


def forward_impl(net, x_bnc, training, skip_last=False, use_plan=True):
    """x (B,N,3) contiguous -> y (B, 3*M) and the tensors backward needs.
    skip_last: stop before fc4 and return None for y -- the caller produces it from saved["zf"][2] / saved["cf"][2] (the
    pair scan of the fused sampler step computes its own queries, fused_step.py).
    use_plan=False: never run on / record a _ForwardPlan (callers that capture the launches into a graph of their own and keep
    the returned tensors alive themselves: surface.py)."""
    convs, fcs = _layers(net)
    B, N, _ = x_bnc.shape
    R = B * N
    rec = None
    if training and FORWARD_PLAN and use_plan:
        capturing = torch.cuda.is_current_stream_capturing()
        plan = _ForwardPlan.acquire(net, x_bnc, skip_last, capturing)
        if plan is not None:
            return plan.run(net, x_bnc)
        if not capturing:  # (a plan's buffers must not come from a graph's private pool)
            rec = []
    saved = {"x": x_bnc, "B": B, "N": N, "zc": [], "cc": [], "zf": [], "cf": [], "training": bool(training)}
    use_batch_stats = training
    a_in, coef_prev = x_bnc.view(R, 3), None
    C5 = convs[-1].Co
    pooled = _empty((B, C5), x_bnc)
    argsel = _empty((B, C5), x_bnc, torch.int32)
    zsel = _empty((B, C5), x_bnc)
    # last conv layer: the max-pool is folded into its epilogue + BatchNorm finalisation when the shapes are 64-aligned
    fuse_pool = training and R > 64 and N % 64 == 0 and C5 % 64 == 0 and convs[-1].Ci % 64 == 0 and FUSE_POOL
    # ... and its BatchNorm finalisation + pool pick into the FC chain when that one runs (B <= 32, the 128 -> 256 x 3 head)
    shape = _fc_chain_shape(fcs[:-1], B) if training and FC_CHAIN and POOL_IN_CHAIN else None
    defer_pool = shape is not None and bool(lib.sn_fc_chain_forward_pool_supported(B, N, *shape))
    if fuse_pool and FX_STATS and _conv_stack_fx(net, convs, x_bnc, B, N, saved, pooled, argsel, zsel, defer_pool, rec):
        convs = []  # the whole stack ran as one call (fixed-point statistics chain)
    for li, L in enumerate(convs):
        if training and fuse_pool and li == len(convs) - 1:
            z, coef = _layer_fwd_bn_pool(R, N, L, a_in, coef_prev, pooled, argsel, zsel)
        elif training:
            z, coef = _layer_fwd_bn(R, L, a_in, coef_prev)
        else:
            z, stats, nblk = _linear_fwd(R, L, a_in, coef_prev, use_batch_stats)
            coef = _bn_coef(L, R, stats, nblk, training)
        saved["zc"].append(z)
        saved["cc"].append(coef)
        a_in, coef_prev = z, coef
    if not fuse_pool:
        check(lib.sn_pool_forward(B, N, C5, ptr(a_in), ptr(coef_prev), ptr(pooled), ptr(argsel), ptr(zsel), _st(x_bnc)),
              "sn_pool_forward")
    saved.update(pooled=pooled, argsel=argsel, zsel=zsel)
    a_in, coef_prev = pooled, None
    hidden = fcs[:-1]
    ob = out_bn(net)
    if training and FC_CHAIN and _fc_chain_fwd(net, hidden, pooled, B, saved, rec, (fcs[-1], ob) if ob is not None else None):
        a_in, coef_prev = saved["zf"][-1], saved["cf"][-1]
        hidden = []
    for L in hidden:
        if L.bn is None:  # ReLU layer without BatchNorm
            z, _, _ = _linear_fwd(B, L, a_in, coef_prev, False, fc_rows=True)
            coef = _identity_coef(L.Co, z)
        elif training and B > 32 and not (B <= 64 and L.Ci in (64, 128, 256)):
            # (33 .. 64 rows: sn_layer_forward_bn keeps both 32-row halves in one workgroup and finalises the BatchNorm in its epilogue)
            # the GEMM, then two-pass batch statistics from z itself (above 32 rows the statistics are not complete inside one
            # workgroup, and sum / sum-of-squares partials lose digits on the head's nearly-constant features)
            z, _, _ = _linear_fwd(B, L, a_in, coef_prev, False, fc_rows=True)
            coef = _empty((4, L.Co), z)
            bn, upd = L.bn, L.bn.track_running_stats
            check(lib.sn_bn_batch_stats_twopass(B, L.Co, ptr(z), ptr(bn.weight), ptr(bn.bias), float(bn.eps), _momentum(bn),
                                                ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None,
                                                ptr(bn.num_batches_tracked) if upd else None, ptr(coef), _st(z)),
                  "sn_bn_batch_stats_twopass")
        elif training:
            z, coef = _layer_fwd_bn(B, L, a_in, coef_prev)
        else:
            z, stats, nblk = _linear_fwd(B, L, a_in, coef_prev, use_batch_stats)
            coef = _bn_coef(L, B, stats, nblk, training)
        saved["zf"].append(z)
        saved["cf"].append(coef)
        a_in, coef_prev = z, coef
    y = saved.get("y_out")  # (the chain's output stage produced the head's output already)
    if y is None and (ob is not None or not skip_last):  # (an output BatchNorm needs every cloud's row: the caller cannot produce y itself)
        y = _last_layer(B, fcs[-1], ob, a_in, coef_prev, training, saved)
    if rec is not None and len(rec) == 2 and not convs and not hidden:
        # the whole head ran as the two fused calls (+ the last layer): from now on steps of this shape replay them
        _ForwardPlan.register(net, x_bnc, skip_last, rec, saved, (fcs[-1], ob))
    return y, saved


def _last_layer(B, L, ob, a_in, coef_prev, training, saved):
    """The head's output layer: y (B, Co) = act(a_in) W^T + b, and -- classification sampler -- the BatchNorm without activation
    behind it (ob = out_bn(net)): one launch up to 32 clouds (sn_layer_forward_bn_out), GEMM + two-pass statistics + apply above.
    Leaves saved["z_out"] / saved["c_out"] (pre-BN output, coefficients) for backward_impl."""
    if ob is None:
        return _linear_fwd(B, L, a_in, coef_prev, False, fc_rows=True)[0]
    bn = ob[1]
    batch_stats = bool(training or not bn.track_running_stats)
    upd = bool(training and bn.track_running_stats)
    coef = _empty((4, L.Co), a_in)
    y = _empty((B, L.Co), a_in)
    rm, rv, nbt = (ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked)) if upd else (None, None, None)
    if batch_stats and B <= 32 and L.Ci in (64, 128, 256, 512):
        z = _empty((B, L.Co), a_in)
        check(lib.sn_layer_forward_bn_out(B, L.Ci, L.Co, ptr(a_in), ptr(coef_prev), ptr(L.W), ptr(L.b), ptr(z), ptr(bn.weight),
                                          ptr(bn.bias), float(bn.eps), _momentum(bn), rm, rv, nbt, ptr(coef), ptr(y), _st(a_in)),
              "sn_layer_forward_bn_out")
    else:
        z = _linear_fwd(B, L, a_in, coef_prev, False, fc_rows=True)[0]
        if not batch_stats:
            rm, rv = ptr(bn.running_mean), ptr(bn.running_var)
        check(lib.sn_bn_output_forward(B, L.Co, 1 if batch_stats else 0, ptr(z), ptr(bn.weight), ptr(bn.bias), float(bn.eps),
                                       _momentum(bn) if batch_stats else 0.0, rm, rv, nbt, ptr(coef), ptr(y), _st(a_in)),
              "sn_bn_output_forward")
    saved["z_out"], saved["c_out"], saved["out_fixed"] = z, coef, not batch_stats
    return y


class _Lease:
    """Held by the `saved` record of a step that runs on a plan's buffers: the plan is free again when the record dies (the
    autograd node that owns it was released, or a no-grad caller dropped it)."""

    __slots__ = ("plan",)

    def __init__(self, plan):
        self.plan = plan
        plan.busy = True

    def __del__(self):
        self.plan.busy = False


class _ForwardPlan:
    """The training forward of the head at one shape as a REPLAY: the C calls of the fused route (conv stack as one call, FC
    chain with the pool stage, last layer) were recorded once together with the tensors they write; later steps of the same
    shape re-issue them with the same argument arrays on the same buffers -- no per-step allocations, no rebuilding of ~25
    pointer arrays (the eager module surface is host-bound: DESIGN.md 6).  Only the cloud pointer, the head's output (a fresh
    tensor: the caller owns it) and the stream change per step.  A plan serves one step at a time: while a step's `saved`
    record is alive (its backward has not run / its graph has not been released) the next forward of that shape records a
    plan of its own (two sampler passes under one loss, main.py:516-524); at most kMaxPlans are kept per module.
    Validity: the parameter / buffer tensors named by the recorded pointer arrays must still be the module's (identity and
    data pointer are checked per step; SampleNet._apply drops the plans when the module is moved or cast).

    A plan that a stream CAPTURE ran on has its buffer addresses baked into somebody's graph (engine.SamplerTrainStep, bench's
    graph legs, a user's own torch.cuda.graph around the step): from then on it is `captured` -- never evicted, never dropped
    as stale, kept alive for the life of the MODULE (net._sn_pinned, which survives .to() / dropped plan lists: a replay of that
    graph must not touch freed memory whatever happens to the module's own list; a graph cannot outlive the module it reads its
    parameters from, so nothing needs the buffers beyond that -- a process that builds many engines on fresh modules gets
    the memory back with them: ADVICE r4), and never handed to an EAGER forward again (an eager step's saved activations would be
    overwritten by the next replay of the graph); later captures may share it (the ring of graphs of one engine runs them one
    after the other)."""

    kMaxPlans = 4
    __slots__ = ("key", "busy", "calls", "saved", "sig", "last", "captured")

    @staticmethod
    def _signature(net):
        convs, fcs = _layers(net)
        sig = []
        for L in convs + fcs:
            sig += [L.W.data_ptr(), L.b.data_ptr()]
            if L.bn is not None:
                bn = L.bn
                sig += [bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                        bn.num_batches_tracked.data_ptr(), bn.eps, bn.momentum]
        ob = out_bn(net)
        if ob is not None:
            bn = ob[1]
            sig += [bn.weight.data_ptr(), bn.bias.data_ptr(), bn.eps, bn.momentum, bn.track_running_stats] + (
                [bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()] if bn.track_running_stats else [])
        return tuple(sig)

    @staticmethod
    def acquire(net, x, skip_last, capturing=False):
        plans = net.__dict__.get("_sn_plans")
        if not plans:
            return None
        key = (x.shape[0], x.shape[1], x.device, bool(skip_last))
        sig = None
        for plan in plans:
            if plan.key == key and not plan.busy and (capturing or not plan.captured):
                if sig is None:
                    sig = _ForwardPlan._signature(net)
                if plan.sig == sig:
                    if capturing and not plan.captured:
                        plan.captured = True
                        net.__dict__.setdefault("_sn_pinned", []).append(plan)
                    return plan
        if sig is not None:  # stale plans of this shape (parameters were replaced): drop them (a captured one only leaves the list)
            net.__dict__["_sn_plans"] = [q for q in plans if q.sig == sig or q.key != key]
        return None

    @staticmethod
    def register(net, x, skip_last, rec, saved, last_layer):
        plans = net.__dict__.setdefault("_sn_plans", [])
        if len([q for q in plans if not q.captured]) >= _ForwardPlan.kMaxPlans:
            free = [q for q in plans if not q.busy and not q.captured]
            if not free:
                return
            plans.remove(free[0])
        plan = _ForwardPlan()
        plan.captured = False
        plan.key = (x.shape[0], x.shape[1], x.device, bool(skip_last))
        plan.sig = _ForwardPlan._signature(net)
        plan.calls = rec
        plan.saved = dict(saved)
        plan.saved.pop("_lease", None)
        plan.saved["_bwd_cache"] = saved["_bwd_cache"] = {}  # static pieces of the backward's argument lists (_conv_stack_bwd_fx)
        # (fcs[-1], out_bn record); None: the caller computes the output layer itself, or the chain's output stage did
        plan.last = None if ((skip_last and last_layer[1] is None) or "y_out" in saved) else last_layer
        plan.busy = False
        saved["_lease"] = _Lease(plan)  # the recording step itself runs on these buffers
        plans.append(plan)

    def run(self, net, x):
        st = _st(x)
        saved = dict(self.saved)
        saved["x"] = x
        saved["zc"], saved["cc"] = list(saved["zc"]), list(saved["cc"])
        saved["_lease"] = _Lease(self)
        xp = x.data_ptr()
        for fn, name, args, xpos, acc, _keep in self.calls:
            if xpos is not None:
                args[xpos] = xp
            args[-1] = st
            rc = fn(*args)
            if rc != 0:
                acc.zero_()
                check(rc, name)
        y = None
        if self.last is not None:
            y = _last_layer(saved["B"], self.last[0], self.last[1], saved["zf"][-1], saved["cf"][-1], True, saved)
        elif "y_out" in saved:
            # the chain's output stage wrote the plan's own buffer: an eager caller receives a copy (the head's output is handed
            # to the script as the simplified cloud, which it may keep across steps); a capture takes the buffer itself (its
            # graph's outputs are static tensors anyway: one launch less per replay)
            y = saved["y_out"] if torch.cuda.is_current_stream_capturing() else saved["y_out"].clone()
        return y, saved


class GradSink(dict):
    """name -> preallocated gradient tensor (views of a flat all-reduce bucket, parallel.FlatGradAllReducer) that the
    backward kernels write into directly, with torch's .grad semantics kept intact:

      * the first backward after reset() -- or after the parameters' .grad were set to None (optimizer.zero_grad(), whose
        default is set_to_none=True) -- OVERWRITES the views: no zero fill, no accumulate kernel;
      * any further backward before the next reset (two sampler forwards under one loss, registration/main.py:516-524;
        gradient accumulation over micro-batches; a retained graph) ACCUMULATES: the kernels write fresh tensors that are
        added to the views;
      * afterwards every parameter's .grad IS its view again (re-bound if zero_grad() had dropped it), so optimizer.step()
        sees the gradients and the bucket stays the single all-reduce operand.
    """

    def __init__(self, views, params):
        super().__init__(views)
        self.params = params  # name -> nn.Parameter
        self.written = False
        self._reducer_ref = None  # weak reference to the FlatGradAllReducer that owns the views (parallel.py)

    @property
    def reducer(self):
        return self._reducer_ref() if self._reducer_ref is not None else None

    def reset(self):
        self.written = False

    def direct(self):
        """True: this backward may write into the views (nothing to preserve in them)."""
        return (not self.written) or all(self.params[n].grad is None for n in self)

    def commit(self, fresh=None):
        """After the kernels ran.  fresh: name -> gradient tensor when the backward could not write in place."""
        for n, view in self.items():
            p = self.params[n]
            mine = p.grad is not None and p.grad.data_ptr() == view.data_ptr() and p.grad.shape == view.shape
            if fresh is not None:
                if p.grad is None:
                    view.copy_(fresh[n])
                elif mine:
                    view.add_(fresh[n])
                else:  # somebody assigned another tensor as .grad: fold it in, then take the slot back
                    torch.add(p.grad, fresh[n], out=view)
            elif p.grad is not None and not mine:
                view.add_(p.grad)
            if not mine:
                p.grad = view
        self.written = True
        red = getattr(self, "reducer", None)
        if red is not None:
            red._unreduced = True  # (a local contribution: a collective captured in a surface graph of this step does not cover it)


def sink_for_backward(net):
    """-> (sink the kernels may write into | None, GradSink to commit afterwards | None)."""
    sink = getattr(net, "_grad_sink", None)
    if sink is None:
        return None, None
    return (sink if sink.direct() else None), sink


def _out(sink, name, like):
    """Gradient destination: the caller-provided sink tensor (e.g. a view of a flat all-reduce bucket) or a fresh one."""
    if sink is not None and name in sink:
        return sink[name]
    return torch.empty_like(like)


# Test hooks (no environment switches; the defaults are the product path, False = the route shapes outside the fast paths'
# support take anyway).  FX_STATS: conv stack as one call with BatchNorm statistics as fixed-point atomics; IN3_CLOSED_FORM:
# conv1's weight gradient in closed form out of conv2's backward; FUSE_POOL: max-pool folded into the last conv layer.
FX_STATS = True
IN3_CLOSED_FORM = True
FUSE_POOL = True
Z1_FREE = True  # one-call conv stack: the xyz layer's activation tensor is not materialised (rebuilt from the cloud where needed)
FC_CHAIN = True  # the FC head's hidden layers as one launch with in-kernel hand-offs (sn_fc_chain_forward)
POOL_IN_CHAIN = True  # ... with the last conv BatchNorm + max-pool pick as its first stage (sn_fc_chain_forward_pool)
FORWARD_PLAN = True  # steps of a shape seen before replay the recorded C calls on recycled buffers (_ForwardPlan)


def _wgrad(R, L, mode, dy, z, kcoef, gsel, argsel, npts, aprev, coef_prev, with_bias, sink=None, name=""):
    dW = _out(sink, name + ".weight", L.W)
    db = _out(sink, name + ".bias", L.b) if with_bias else None
    nsplit = lib.sn_linear_wgrad_splits(R, L.Ci, L.Co, 1 if with_bias else 0)
    part = _empty((nsplit * L.Co * (L.Ci + (1 if with_bias else 0)),), L.W)
    check(lib.sn_linear_wgrad(R, L.Ci, L.Co, mode, ptr(dy), ptr(z), ptr(kcoef), ptr(gsel), ptr(argsel), npts, ptr(aprev),
                              ptr(coef_prev), ptr(part), ptr(dW), ptr(db), _st(L.W)), "sn_linear_wgrad")
    return dW, db


def _dgrad(R, L, mode, dy, z, kcoef, gsel, argsel, npts, zprev, coef_prev):
    dyprev = _empty((R, L.Ci), L.W)
    nblk = lib.sn_linear_stats_blocks(R)
    stats = _empty((nblk, 2, L.Ci), L.W) if coef_prev is not None else None
    check(lib.sn_linear_dgrad(R, L.Ci, L.Co, mode, ptr(dy), ptr(z), ptr(kcoef), ptr(gsel), ptr(argsel), npts, ptr(L.W),
                              ptr(zprev), ptr(coef_prev), ptr(dyprev), ptr(stats), _st(L.W)), "sn_linear_dgrad")
    return dyprev, stats, nblk


def _bwd_layer(R, L, mode, dy, z, kcoef, gsel, argsel, npts, zprev, coef_prev, sink=None, name=""):
    """dgrad + wgrad of one layer through sn_linear_backward (one launch on the fast path)."""
    dW = _out(sink, name + ".weight", L.W)
    dyprev = _empty((R, L.Ci), L.W)
    nblk = lib.sn_linear_stats_blocks(R)
    stats = _empty((nblk, 2, L.Ci), L.W)
    nsplit = lib.sn_linear_wgrad_splits(R, L.Ci, L.Co, 0)
    part = _empty((nsplit * L.Co * L.Ci,), L.W)
    check(lib.sn_linear_backward(R, L.Ci, L.Co, mode, ptr(dy), ptr(z), ptr(kcoef), ptr(gsel), ptr(argsel), npts, ptr(L.W),
                                 ptr(zprev), ptr(coef_prev), ptr(dyprev), ptr(stats), ptr(part), ptr(dW), _st(L.W)),
          "sn_linear_backward")
    return dW, dyprev, stats, nblk


def _bn_bwd(L, R, stats, nblk, coef, sink=None, bn_name="", lin_name=""):
    C = L.Co
    dgamma, dbeta = _out(sink, bn_name + ".weight", L.bn.weight), _out(sink, bn_name + ".bias", L.bn.bias)
    dbias = _out(sink, lin_name + ".bias", L.b)
    kcoef = _empty((3, C), L.W)
    check(lib.sn_bn_backward_coef(nblk, C, R, ptr(stats), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(kcoef),
                                  _st(L.W)), "sn_bn_backward_coef")
    return dgamma, dbeta, dbias, kcoef


def _layer_bwd(R, L, mode, dy, z, kcoef, gsel, argsel, npts, zprev, coef_prev, Lprev, sink, name, bn_prev, lin_prev,
               with_bias, prev_bn_rows=0):
    """Backward of layer `name`: dW (db when with_bias), dYprev and -- when the layer below (Lprev) has a BatchNorm --
    that BatchNorm's dgamma / dbeta, the bias gradient of the layer below, and its dZ coefficients (kcoef)."""
    dW = _out(sink, name + ".weight", L.W)
    db = _out(sink, name + ".bias", L.b) if with_bias else None
    dyprev = _empty((R, L.Ci), L.W)
    has_bn = coef_prev is not None
    stats = _empty((lib.sn_linear_stats_blocks(R), 2, L.Ci), L.W) if has_bn else None
    nsplit = lib.sn_linear_wgrad_splits(R, L.Ci, L.Co, 1 if with_bias else 0)
    part = _empty((nsplit * L.Co * (L.Ci + (1 if with_bias else 0)),), L.W)
    dg = dbt = dbs = kc = None
    if has_bn:
        if Lprev.bn is not None:
            dg, dbt = _out(sink, bn_prev + ".weight", Lprev.bn.weight), _out(sink, bn_prev + ".bias", Lprev.bn.bias)
        else:  # identity coefficients (no BatchNorm below): only the bias gradient of the layer below is meaningful
            dg, dbt = _empty((L.Ci,), L.W), _empty((L.Ci,), L.W)
            prev_bn_rows = -1
        dbs = _out(sink, lin_prev + ".bias", Lprev.b)
        kc = _empty((3, L.Ci), L.W)
    check(lib.sn_layer_backward(R, L.Ci, L.Co, mode, ptr(dy), ptr(z), ptr(kcoef), ptr(gsel), ptr(argsel), npts, ptr(L.W),
                                ptr(zprev), ptr(coef_prev), ptr(dyprev), ptr(stats), ptr(part), ptr(dW), ptr(db), ptr(dg),
                                ptr(dbt), ptr(dbs), ptr(kc), int(prev_bn_rows), _st(L.W)), "sn_layer_backward")
    return dW, db, dyprev, dg, dbt, dbs, kc


def conv_stack_backward_supported(net, B, N):
    """True when backward_impl will run the conv stack through sn_conv_stack_backward (whose closing kernel can carry the
    loss side's deferred tail, fused_step.py)."""
    import ctypes

    if not (FX_STATS and IN3_CLOSED_FORM):
        return False
    convs, _ = _layers(net)
    chans = (ctypes.c_int * (len(convs) + 1))(convs[0].Ci, *[L.Co for L in convs])
    return lib.sn_conv_stack_backward_scratch_floats(B, N, len(convs), chans) > 0


def _conv_stack_bwd_fx(net, convs, saved, gsel, kcoef_top, sink, grads, names_c, bn_c, step_tail=None):
    """Backward of the conv stack as one call (sn_conv_stack_backward: 5 launches, BatchNorm-backward sums as fixed-point
    atomics, every weight-gradient partial reduced by the closing kernel).  Returns False when the shapes are not
    supported (the per-layer path runs instead)."""
    import ctypes

    n = len(convs)
    B, N = saved["B"], saved["N"]
    x = saved["x"]
    VP = ctypes.c_void_p * n

    def arr(ts):
        return VP(*[ptr(t) for t in ts])

    # what does not change from step to step when the forward ran on a plan's buffers (saved["_bwd_cache"], shared by the
    # steps of that plan): shape arrays, scratch, the pointer arrays of the weights and of the saved activations
    cache = saved.get("_bwd_cache")
    st = cache.get("conv") if cache is not None else None
    if st is None:
        chans = (ctypes.c_int * (n + 1))(convs[0].Ci, *[L.Co for L in convs])
        nscr = lib.sn_conv_stack_backward_scratch_floats(B, N, n, chans)
        if nscr <= 0:
            return False
        acc = getattr(net, "_fx_acc_b", None)
        nacc = lib.sn_conv_stack_acc_elems(n)
        if acc is None or acc.device != x.device or acc.numel() != nacc:
            acc = torch.zeros(nacc, device=x.device, dtype=torch.int64)  # persistent: every call leaves it zero
            net._fx_acc_b = acc
        scratch = _empty((nscr,), x)
        st = (chans, acc, scratch, arr([L.W for L in convs]), arr(saved["zc"]), arr(saved["cc"]))
        if cache is not None:
            cache["conv"] = st
    chans, acc, scratch, W_arr, zc_arr, cc_arr = st
    dW = [_out(sink, names_c[i] + ".weight", convs[i].W) for i in range(n)]
    dg = [_out(sink, bn_c[i] + ".weight", convs[i].bn.weight) for i in range(n - 1)]
    dbt = [_out(sink, bn_c[i] + ".bias", convs[i].bn.bias) for i in range(n - 1)]
    dbs = [_out(sink, names_c[i] + ".bias", convs[i].b) for i in range(n - 1)]
    try:
        check(lib.sn_conv_stack_backward(B, N, n, chans, ptr(x), W_arr, ptr(convs[0].b), zc_arr, cc_arr, ptr(gsel),
                                         ptr(saved["argsel"]), ptr(kcoef_top), ptr(acc), ptr(scratch),
                                         arr(dW), arr(dg + [None]), arr(dbt + [None]), arr(dbs + [None]), step_tail, _st(x)),
              "sn_conv_stack_backward")
    except Exception:
        acc.zero_()
        raise
    for i in range(n):
        grads[names_c[i] + ".weight"] = dW[i]
    for i in range(n - 1):
        grads[bn_c[i] + ".weight"], grads[bn_c[i] + ".bias"], grads[names_c[i] + ".bias"] = dg[i], dbt[i], dbs[i]
    return True


def _fc_chain_bwd(net, convs, fcs, saved, grad_y, sink, grads, fixed, obn=None):
    """The FC head's backward as one launch (sn_fc_chain_backward).  Fills `grads` for every FC parameter and for the last
    conv layer's BatchNorm / bias; returns (gsel, kcoef_top) for the conv stack's backward, or False when the shape is not
    supported (the per-layer launches run instead)."""
    import ctypes

    nf = len(fcs)
    B, R = saved["B"], saved["B"] * saved["N"]
    zf, cf, cc = saved["zf"], saved["cf"], saved["cc"]
    like = grad_y
    L5 = convs[-1]
    C5 = L5.Co
    VP = ctypes.c_void_p * nf

    def arr(ts):
        return VP(*[ptr(t) for t in ts])

    cache = saved.get("_bwd_cache")
    st = cache.get(("fc", fixed)) if cache is not None else None
    if st is None:
        Co = (ctypes.c_int * nf)(*[fcs[j].Co for j in range(nf - 1, -1, -1)])
        Ci = (ctypes.c_int * nf)(*[fcs[j].Ci for j in range(nf - 1, -1, -1)])
        if nf > 5 or not lib.sn_fc_chain_backward_supported(B, nf, Co, Ci):
            return False
        sync = getattr(net, "_fc_sync_b", None)
        if sync is None or sync.device != like.device:
            sync = torch.zeros(16, SYNC_STRIDE, device=like.device, dtype=torch.int32)  # persistent: epoch + monotonic arrival counters
            net._fc_sync_b = sync
        st = {"Co": Co, "Ci": Ci, "sync": sync, "xbuf": _empty((nf * 32 * 256,), like), "gsel": _empty((B, C5), like),
              "kcoef": _empty((3, C5), like)}
        if cache is not None:
            cache[("fc", fixed)] = st
    Co, Ci, sync, xbuf, gsel, kcoef = st["Co"], st["Ci"], st["sync"], st["xbuf"], st["gsel"], st["kcoef"]
    W, zprev, coefprev, rows, dg, dbt, dbs, dW, aprev, araw = [], [], [], [], [], [], [], [], [], []
    keep = []
    for j in range(nf - 1, -1, -1):
        L = fcs[j]
        W.append(L.W)
        dW.append(_out(sink, L.name + ".weight", L.W))
        grads[L.name + ".weight"] = dW[-1]
        if j > 0:
            Lp = fcs[j - 1]
            zprev.append(zf[j - 1]), coefprev.append(cf[j - 1]), aprev.append(zf[j - 1]), araw.append(0)
            rows.append(-1 if (fixed or Lp.bn is None) else B)
            if Lp.bn is not None:
                dg.append(_out(sink, Lp.bn_name + ".weight", Lp.bn.weight)), dbt.append(_out(sink, Lp.bn_name + ".bias", Lp.bn.bias))
                grads[Lp.bn_name + ".weight"], grads[Lp.bn_name + ".bias"] = dg[-1], dbt[-1]
            else:  # identity coefficients: nothing to learn there
                dg.append(_empty((Lp.Co,), like)), dbt.append(_empty((Lp.Co,), like))
                keep += [dg[-1], dbt[-1]]
            dbs.append(_out(sink, Lp.name + ".bias", Lp.b))
            grads[Lp.name + ".bias"] = dbs[-1]
        else:  # fc1 sits on the max-pool: the layer "below" is the last conv layer seen through the selected points
            zprev.append(saved["zsel"]), coefprev.append(cc[-1]), aprev.append(saved["pooled"]), araw.append(1)
            rows.append(-1 if fixed else R)
            dg.append(_out(sink, L5.bn_name + ".weight", L5.bn.weight)), dbt.append(_out(sink, L5.bn_name + ".bias", L5.bn.bias))
            dbs.append(_out(sink, L5.name + ".bias", L5.b))
            grads[L5.bn_name + ".weight"], grads[L5.bn_name + ".bias"], grads[L5.name + ".bias"] = dg[-1], dbt[-1], dbs[-1]
    db_top = _out(sink, fcs[-1].name + ".bias", fcs[-1].b)
    grads[fcs[-1].name + ".bias"] = db_top
    ins = st.get("ins")
    if ins is None:  # (pointer arrays of the operands: static with the plan's buffers)
        ins = (arr(W), arr(zprev), arr(coefprev), (ctypes.c_longlong * nf)(*rows), arr(aprev), (ctypes.c_int * nf)(*araw))
        st["ins"] = ins
    if obn is not None:  # the output BatchNorm's backward opens the launch (obn: z, coef, fixed, dgamma, dbeta)
        check(lib.sn_fc_chain_backward_obn(B, nf, Co, Ci, ptr(grad_y), ptr(obn[0]), ptr(obn[1]), int(obn[2]), ptr(obn[3]), ptr(obn[4]),
                                           ins[0], ins[1], ins[2], ins[3], arr(dg), arr(dbt), arr(dbs), arr(dW), ptr(db_top), ins[4],
                                           ins[5], ptr(gsel), ptr(kcoef), ptr(xbuf), ptr(sync), _st(like)), "sn_fc_chain_backward_obn")
    else:
        check(lib.sn_fc_chain_backward(B, nf, Co, Ci, ptr(grad_y), ins[0], ins[1], ins[2], ins[3],
                                       arr(dg), arr(dbt), arr(dbs), arr(dW), ptr(db_top), ins[4], ins[5],
                                       ptr(gsel), ptr(kcoef), ptr(xbuf), ptr(sync), _st(like)), "sn_fc_chain_backward")
    saved["fc_chain_b"] = (xbuf, keep)  # (scratch of the asynchronous launch)
    return gsel, kcoef


def backward_impl(net, saved, grad_y, sink=None, after_fc=None, step_tail=None):
    """-> dict parameter-name -> gradient tensor (every parameter of the MLP).
    sink: optional dict name -> preallocated tensor the gradient is written into (overwritten, not accumulated).
    after_fc: optional callback invoked once all FC-head gradients have been enqueued (DP overlap point)."""
    convs, fcs = _layers(net)
    names_c, bn_c = [L.name for L in convs], [L.bn_name for L in convs]
    names_f, bn_f = [L.name for L in fcs], [L.bn_name for L in fcs]
    nf = len(fcs)
    B, N = saved["B"], saved["N"]
    R = B * N
    grads = {}
    grad_y = grad_y.contiguous()
    zf, cf, zc, cc = saved["zf"], saved["cf"], saved["zc"], saved["cc"]

    def _z1():
        # the xyz layer's pre-BatchNorm output for the per-layer backward when the forward did not keep it (Z1_FREE)
        if zc[0] is None:
            zc[0] = _linear_fwd(R, convs[0], saved["x"].view(R, 3), None, False)[0]
        return zc[0]

    # eval-mode forward (running statistics): every BatchNorm backward is dZ = scale * dY -- the kernels take "rows < 0" for
    # that (no batch-statistics terms); the fixed-point / closed-form fast paths assume batch statistics and are skipped
    fixed = not saved.get("training", True)
    bn_rows = -1 if fixed else 0

    obn_args = None
    if "z_out" in saved:
        # the BatchNorm (no activation) behind the last FC layer: dZ of that layer, the BatchNorm's own gradients -- as the
        # opening of the FC chain's backward launch where that one runs, else a launch of its own
        obn, ob = out_bn(net)
        zo = saved["z_out"]
        dgo, dbo = _out(sink, obn + ".weight", ob.weight), _out(sink, obn + ".bias", ob.bias)
        grads[obn + ".weight"], grads[obn + ".bias"] = dgo, dbo
        obn_args = (zo, saved["c_out"], saved["out_fixed"], dgo, dbo)

    # ---- FC head (rows = B): fc4 -> fc3 -> fc2 -> fc1 -> pooled features ----
    chain = FC_CHAIN and B <= 32 and _fc_chain_bwd(net, convs, fcs, saved, grad_y, sink, grads, fixed, obn_args)
    if obn_args is not None and not chain:
        dz = _empty(tuple(zo.shape), zo)
        check(lib.sn_bn_output_backward(B, zo.shape[1], 1 if saved["out_fixed"] else 0, ptr(grad_y), ptr(zo), ptr(saved["c_out"]),
                                        ptr(dz), ptr(dgo), ptr(dbo), _st(zo)), "sn_bn_output_backward")
        grad_y = dz
    dy, kcoef = grad_y, None
    if chain:
        dy, kcoef = chain
    for j in (range(nf - 1, -1, -1) if not chain else ()):
        L = fcs[j]
        mode = DZ_PLAIN if j == nf - 1 else DZ_BN
        if j > 0:
            zprev, cprev, Lprev, bnp, linp, rows = zf[j - 1], cf[j - 1], fcs[j - 1], bn_f[j - 1], names_f[j - 1], bn_rows
        elif B > 32:  # the epilogue trick needs the register-resident (R <= 32) kernels: separate pooling backward below
            zprev, cprev, Lprev, bnp, linp, rows = saved["pooled"], None, None, None, None, 0
        else:
            # fc1 sits on the max-pool: its "previous layer" is conv5 seen through the selected points -- the ReLU mask and
            # BatchNorm-backward sums of the dgrad epilogue over zsel ARE the pooling backward (no separate launch)
            zprev, cprev, Lprev, bnp, linp, rows = saved["zsel"], cc[4], convs[4], bn_c[4], names_c[4], (-1 if fixed else R)
        dW, db, dy, dg, dbt, dbs, kc = _layer_bwd(B, L, mode, dy, zf[j] if j < nf - 1 else None, kcoef, None, None, 1, zprev, cprev,
                                                  Lprev, sink, names_f[j], bnp, linp, j == nf - 1, rows)
        grads[names_f[j] + ".weight"] = dW
        if db is not None:
            grads[names_f[j] + ".bias"] = db
        if linp is not None:
            grads[linp + ".bias"] = dbs
            if Lprev.bn is not None:
                grads[bnp + ".weight"], grads[bnp + ".bias"] = dg, dbt
        kcoef = kc
    gsel = dy  # (B, C5): gradient at the selected (max-pooled) points, already masked; kcoef = conv5's BatchNorm backward
    if after_fc is not None:
        after_fc()
    if B > 32:  # dy is the gradient w.r.t. the pooled features: max-pool + conv5's BatchNorm backward in their own launch
        C5, L5, g_pool = convs[4].Co, convs[4], dy
        gsel = _empty((B, C5), grad_y)
        dgamma, dbeta = _out(sink, bn_c[4] + ".weight", L5.bn.weight), _out(sink, bn_c[4] + ".bias", L5.bn.bias)
        dbias = _out(sink, names_c[4] + ".bias", L5.b)
        kcoef = _empty((3, C5), grad_y)
        check(lib.sn_pool_backward_bn(B, C5, -1 if fixed else R, ptr(g_pool), ptr(saved["pooled"]), ptr(saved["zsel"]), ptr(gsel), ptr(cc[4]),
                                      ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(kcoef), _st(grad_y)), "sn_pool_backward_bn")
        grads[bn_c[4] + ".weight"], grads[bn_c[4] + ".bias"], grads[names_c[4] + ".bias"] = dgamma, dbeta, dbias

    # ---- conv stack (rows = B*N): conv5 -> ... -> conv2 (each also finishes the BatchNorm of the layer below), conv1 ----
    if step_tail is not None:  # the loss value the deferred tail writes turns NaN when a chain launch of this step timed out
        check(lib.sn_step_tail_set_error_words(step_tail, ptr(getattr(net, "_fc_sync", None)), ptr(getattr(net, "_fc_sync_b", None))),
              "sn_step_tail_set_error_words")
    if not fixed and FX_STATS and IN3_CLOSED_FORM and _conv_stack_bwd_fx(net, convs, saved, gsel, kcoef, sink, grads, names_c, bn_c, step_tail):
        return grads
    if step_tail is not None:
        raise RuntimeError("backward_impl: a deferred step tail needs the one-call conv stack backward (conv_stack_backward_supported)")
    dy = None
    in3_floats = lib.sn_layer_backward_in3_stats_floats(R, convs[1].Ci, convs[1].Co) if (IN3_CLOSED_FORM and convs[0].Ci == 3 and not fixed) else 0
    for i in (4, 3, 2, 1):
        L = convs[i]
        if i == 1 and in3_floats > 0:
            # conv2 sits on the xyz input layer: its fused backward also yields conv1's weight gradient (closed form from
            # three extra per-channel sums + the moments of x: no pass of its own over dY1)
            Lp = convs[0]
            dW = _out(sink, names_c[1] + ".weight", L.W)
            dW0 = _out(sink, names_c[0] + ".weight", Lp.W)
            dg, dbt = _out(sink, bn_c[0] + ".weight", Lp.bn.weight), _out(sink, bn_c[0] + ".bias", Lp.bn.bias)
            dbs = _out(sink, names_c[0] + ".bias", Lp.b)
            stats = _empty((in3_floats,), L.W)
            part = _empty((lib.sn_linear_wgrad_splits(R, L.Ci, L.Co, 0) * L.Co * L.Ci,), L.W)
            kc = _empty((3, L.Ci), L.W)
            check(lib.sn_layer_backward_in3(R, L.Ci, L.Co, ptr(dy), ptr(zc[1]), ptr(kcoef), ptr(L.W), ptr(_z1()), ptr(cc[0]),
                                            ptr(stats), ptr(part), ptr(dW), ptr(dg), ptr(dbt), ptr(dbs), ptr(kc),
                                            ptr(saved["x"]), ptr(Lp.W), ptr(Lp.b), ptr(dW0), _st(L.W)), "sn_layer_backward_in3")
            grads[names_c[1] + ".weight"], grads[names_c[0] + ".weight"] = dW, dW0
            grads[bn_c[0] + ".weight"], grads[bn_c[0] + ".bias"], grads[names_c[0] + ".bias"] = dg, dbt, dbs
            break
        mode = DZ_POOL if i == 4 else DZ_BN
        gs, ag = (gsel, saved["argsel"]) if i == 4 else (None, None)
        dW, _, dy, dg, dbt, dbs, kc = _layer_bwd(R, L, mode, dy, zc[i], kcoef, gs, ag, N, zc[i - 1] if i > 1 else _z1(), cc[i - 1], convs[i - 1],
                                                 sink, names_c[i], bn_c[i - 1], names_c[i - 1], False, bn_rows)
        grads[names_c[i] + ".weight"] = dW
        grads[bn_c[i - 1] + ".weight"], grads[bn_c[i - 1] + ".bias"], grads[names_c[i - 1] + ".bias"] = dg, dbt, dbs
        kcoef = kc
    else:
        dW, _ = _wgrad(R, convs[0], DZ_BN, dy, _z1(), kcoef, None, None, N, saved["x"].view(R, 3), None, False, sink, names_c[0])
        grads[names_c[0] + ".weight"] = dW
    return grads


class PointNetMLPFunction(torch.autograd.Function):
    """y (B, 3M) = head(x (B,N,3)); differentiable w.r.t. all 34 parameter tensors (36 with an output BatchNorm; x is data: no
    gradient)."""

    @staticmethod
    def forward(ctx, net, x_bnc, training, *params):
        with torch.cuda.device(x_bnc.device):
            y, saved = forward_impl(net, x_bnc, training)
        ctx.net, ctx.saved = net, saved
        return y

    @staticmethod
    def backward(ctx, grad_y):
        net = ctx.net
        sink, owner = sink_for_backward(net)
        with torch.cuda.device(grad_y.device):
            grads = backward_impl(net, ctx.saved, grad_y, sink, getattr(net, "_after_fc_grads", None) if sink is not None else None)
            if owner is not None:
                owner.commit(None if sink is not None else grads)
        # (ctx.saved stays: a retained graph may run backward again)
        # gradients that went into the sink are not handed to autograd (nothing left to accumulate)
        return (None, None, None) + tuple(None if (owner is not None and n in owner) else grads[n] for n in param_order(net))


def pointnet_head(net, x_bnc):
    """x (B,N,3) float32 CUDA -> y (B, 3, M) exactly as samplenet.py:90-104 produces it."""
    if not x_bnc.is_cuda:
        raise RuntimeError("samplenet_amd.pointnet runs on the GPU only; no CPU fallback exists")
    if x_bnc.dtype != torch.float32:
        raise TypeError("expected float32")
    x_bnc = x_bnc.contiguous()
    params = param_list(net)
    comm = net.__dict__.get("_sn_sync_bn") if net.training else None
    if comm is not None:
        # batch statistics over all ranks (syncbn.convert_sync_batchnorm): the layer-by-layer route with one collective per BatchNorm
        from . import syncbn

        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            y = syncbn.SyncBNMLPFunction.apply(net, x_bnc, comm, *params)
        else:
            with torch.cuda.device(x_bnc.device):
                y, _ = syncbn.forward_sync(net, x_bnc, comm)
    elif torch.is_grad_enabled() and any(p.requires_grad for p in params):
        y = PointNetMLPFunction.apply(net, x_bnc, net.training, *params)
    else:
        with torch.cuda.device(x_bnc.device):
            y, _ = forward_impl(net, x_bnc, net.training)
    if out_bn(net) is None:
        last_bn = getattr(net, "bn_fc%d" % getattr(net, "num_fc_layers", 4), None)
        if last_bn is not None:  # only torch.nn.SyncBatchNorm gets here (syncbn.convert_sync_batchnorm): statistics over all ranks
            y = last_bn(y)
    return y.view(-1, 3, net.num_out_points)
