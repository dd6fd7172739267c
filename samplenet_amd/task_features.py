"""The registration task network -- PointNetFeatures, PCRNet, its Chamfer loss -- on HIP kernels.

Drop-in for `registration/models/pcrnet.py:8-41` (same constructor, parameter names conv1..conv5 -> state_dict
compatible): five 1x1 convolutions 3 -> 64 -> 64 -> 64 -> 128 -> bottleneck with ReLU and NO BatchNorm, then the max over
the points.  It is the other half of every real sampler training step: PCRNet is frozen, but the gradient of the task
loss reaches the sampler THROUGH it, so forward and the data gradient matter (weight gradients are produced too, for
training the task network itself).

Built on the same C-ABI entries as the sampler's own feature extractor (samplenet_amd/pointnet.py): the fused
`act = relu(scale * z + shift)` operand mode with scale = 1, shift = 0 is exactly ReLU (fma(z, 1, 0) == z), and the
pooling / sparse last-layer gradient use the BatchNorm-free coefficients k = (1, 0, 0).
"""
import weakref

import torch
import torch.nn as nn

from ._lib import check, lib, ptr, stream_of

_DZ_PLAIN, _DZ_POOL = 0, 2


def _st(t):
    return stream_of(t)


# Test hooks (the defaults are the product path; False = the route other shapes take anyway).  The FC trunk and the output head
# have ONE route -- sn_skinny_linear / sn_skinny_wgrad, sn_pcrnet_head_* -- on the GPU, whatever the widths: no torch.nn.Linear.
FUSE_MAXPOOL = True        # last layer + max over the points as one GEMM launch (sn_linear_forward_maxpool)
FUSE_NARROW = True         # conv1..conv4 (3 -> 64 -> 64 -> 64 -> 128) as one launch (sn_pointnet_narrow_forward)
WIDE_MAXPOOL = True        # ... as the wide kernel (A fragments resident, pre-split weight planes) where the shape allows
SPARSE_POOL_DGRAD = True   # last layer's data gradient from the one non-zero per cloud and channel (sn_pool_dgrad_sparse)

_CONST = {}  # (rows, channels, device) -> constant coefficient table, built once (never written afterwards)


def _unit_rows(rows, c, like):
    """(rows, c) table whose row 0 is 1 and whose other rows are 0.  rows = 4: operand coefficients scale = 1, shift = 0, i.e.
    plain ReLU (rows 2, 3 unused); rows = 3: k = (1, 0, 0), dZ = dY.  Constant, so one table per shape and device serves every
    call -- two fill launches per layer and call otherwise (24 per step of the registration loop, 4.4 us each in a graph).
    While a stream capture is under way a missing table is built for that call only (its memory would belong to the graph)."""
    key = (rows, c, like.device)
    t = _CONST.get(key)
    if t is None:
        t = torch.zeros(rows, c, device=like.device, dtype=torch.float32)
        t[0].fill_(1.0)
        if not torch.cuda.is_current_stream_capturing():
            _CONST[key] = t
    return t


_PLANES = {}  # ids of the weight parameters -> (weak references to them, their versions, their bf16 planes)
# Weights declared CONSTANT for captured graphs: under a stream capture the split of such weights is not recorded again when the
# cached planes are those of the current version -- a frozen task network's graphs then carry no split launches (four of 37
# launches of the registration task term).  Two ways in: PCRNet.static_weights(True) (a promise of the caller, for graphs it
# captures itself: engine.SamplerTrainStep(task_loss=...), torch.cuda.graph around a step -- after changing the weights in place,
# recapture), and graphed.py's own graphs (automatic: its guard compares the parameters' version counters before every replay and
# recaptures).  The planes a graph was captured on stay alive with the registry (_STATIC_KEEP).
_STATIC = {}        # id(parameter) -> weak reference to it (ids, not a WeakSet: tensors compare elementwise)
_static_ctx = []    # stack of sets of id(parameter): static for the duration of a capture (graphed.static_capture)
_STATIC_KEEP = {}   # id(planes tensor) -> planes tensor (never dropped: a replayed graph may read them)


def _is_static(bases):
    def declared(b):
        r = _STATIC.get(id(b))
        return r is not None and r() is b

    return all(declared(b) or any(id(b) in ctx for ctx in _static_ctx) for b in bases)


class static_capture:
    """with static_capture(parameters): these parameters are constant for graphs captured inside (see _STATIC)."""

    def __init__(self, params):
        self.ids = {id(p._base if p._base is not None else p) for p in params}

    def __enter__(self):
        _static_ctx.append(self.ids)

    def __exit__(self, *exc):
        _static_ctx.remove(self.ids)


def _weight_planes(*Ws, tag=""):
    """Scratch for the split of the weights Ws into three bf16 planes each (sn_linear_forward_maxpool_wide, sn_pointnet_narrow_forward)
    and whether it already holds the split of these very weights: eager calls reuse it while the parameter objects are the same
    and their version counters stand still (the two clouds of a registration step, every step of a frozen task network); under
    a stream capture the split is recorded again -- a replay must see weights that were updated in place since -- unless the
    weights were declared constant (_STATIC / static_capture above).
    tag: distinguishes the images kept of the same weights (the backward's transposed planes)."""
    bases = [W._base if W._base is not None else W for W in Ws]
    key = (tag,) + tuple(id(b) for b in bases)
    vers = tuple((b._version, b.data_ptr()) for b in bases)  # (a write through .data keeps the version: the storage address is the second witness)
    numel = 3 * sum(W.numel() for W in Ws)
    hit = _PLANES.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if (hit is not None and all(r() is b for r, b in zip(hit[0], bases)) and hit[2].numel() == numel
            and hit[2].device == Ws[0].device):
        static = capturing and hit[1] == vers and _is_static(bases)
        ready = hit[1] == vers and (not capturing or static)
        if static:
            _STATIC_KEEP[id(hit[2])] = hit[2]
        if not capturing:
            _PLANES[key] = (hit[0], vers, hit[2])
        return hit[2], ready
    planes = torch.empty(numel, device=Ws[0].device, dtype=torch.bfloat16)
    if not capturing:
        for k in [k for k, v in _PLANES.items() if any(r() is None for r in v[0])]:
            del _PLANES[k]
        _PLANES[key] = (tuple(weakref.ref(b) for b in bases), vers, planes)
    return planes, False


def _sparse_last(ctx_needs, nl, B, N, Ws):
    """The last layer's backward can take the sparse route: frozen weights of that layer, a layer below it, supported shape."""
    return (SPARSE_POOL_DGRAD and nl > 1 and not ctx_needs[1 + 2 * (nl - 1)] and not ctx_needs[2 + 2 * (nl - 1)]
            and bool(lib.sn_pool_dgrad_sparse_supported(B, N, Ws[-1].shape[1], Ws[-1].shape[0])))


def _ident(c, like):
    return _unit_rows(4, c, like)


class _FeaturesFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_bnc, *wb):
        B, N, _ = x_bnc.shape
        R = B * N
        Ws, bs = wb[0::2], wb[1::2]
        zs, a_in, coef_prev = [], x_bnc.reshape(R, 3), None
        idents = []
        need_grad = any(ctx.needs_input_grad)
        with torch.cuda.device(x_bnc.device):
            st = _st(x_bnc)
            dev = x_bnc.device
            C = Ws[-1].shape[0]
            pooled = torch.empty(B, C, device=dev, dtype=torch.float32)
            argsel = torch.empty(B, C, device=dev, dtype=torch.int32)
            zsel = torch.empty(B, C, device=dev, dtype=torch.float32)
            nl = len(Ws)
            first = 0
            tile_keys = None
            if (FUSE_NARROW and nl == 5 and Ws[0].shape[1] == 3 and all(W.shape[1] == Wp.shape[0] for Wp, W in zip(Ws[:3], Ws[1:4]))
                    and lib.sn_pointnet_narrow_forward_supported(R, *[W.shape[0] for W in Ws[:4]])):
                # conv1..conv4 in one launch; their pre-activations are kept only where a backward will read them
                planes, ready = _weight_planes(Ws[1], Ws[2], Ws[3])
                z123 = [torch.empty(R, 64, device=dev, dtype=torch.float32) if need_grad else None for _ in range(3)]
                z4 = torch.empty(R, 128, device=dev, dtype=torch.float32)
                # (rider: the key scratch of the pooled layer's tile kernel, where that one follows, is cleared by this launch)
                Cl, Cil = Ws[4].shape
                tile_keys = None
                if (FUSE_MAXPOOL and lib.sn_linear_forward_maxpool_supported(R, Cil, Cl, N)
                        and not (WIDE_MAXPOOL and lib.sn_linear_forward_maxpool_wide_supported(R, Cil, Cl, N))):
                    tile_keys = torch.empty(B * 2 * Cl, device=dev, dtype=torch.int64)
                check(lib.sn_pointnet_narrow_forward(R, ptr(a_in), ptr(Ws[0]), ptr(bs[0]), ptr(Ws[1]), ptr(bs[1]), ptr(Ws[2]),
                                                     ptr(bs[2]), ptr(Ws[3]), ptr(bs[3]), ptr(planes), int(ready), ptr(z123[0]),
                                                     ptr(z123[1]), ptr(z123[2]), ptr(z4), ptr(tile_keys),
                                                     tile_keys.numel() if tile_keys is not None else 0, st), "sn_pointnet_narrow_forward")
                zs = z123 + [z4]
                idents = [_ident(W.shape[0], z4) for W in Ws[:4]]
                a_in, coef_prev, first = z4, idents[-1], 4
            for li, (W, b) in enumerate(zip(Ws, bs)):
                if li < first:
                    continue
                Co, Ci = W.shape[0], W.shape[1]
                if li == nl - 1 and li > 0 and FUSE_MAXPOOL and lib.sn_linear_forward_maxpool_supported(R, Ci, Co, N):
                    # last layer + max over the points in one GEMM: its activations are written only when a backward will
                    # read them; the frozen / no-gradient branch (the registration loop's template cloud) never materialises them
                    # (the dense backward of this layer reads them; the sparse one -- frozen weights, <= 64 points -- does not)
                    keep_z = need_grad and not _sparse_last(ctx.needs_input_grad, nl, B, N, Ws)
                    z = torch.empty(R, Co, device=dev, dtype=torch.float32) if keep_z else None
                    if WIDE_MAXPOOL and lib.sn_linear_forward_maxpool_wide_supported(R, Ci, Co, N):
                        planes, ready = _weight_planes(W)
                        nb = lib.sn_linear_forward_maxpool_wide_scratch_bytes(R, Ci, Co, N)
                        keys = torch.empty(nb // 8, device=dev, dtype=torch.int64)
                        # (the rows of the maxima only where a backward will ask for them: the kernel then skips the argmax)
                        check(lib.sn_linear_forward_maxpool_wide(R, Ci, Co, N, ptr(a_in), ptr(coef_prev), ptr(W), ptr(b), ptr(z),
                                                                 ptr(keys), ptr(pooled), ptr(argsel) if need_grad else None,
                                                                 ptr(zsel) if need_grad else None, ptr(planes), int(ready), st),
                              "sn_linear_forward_maxpool_wide")
                    else:
                        keys = tile_keys if tile_keys is not None else torch.empty(B * 2 * Co, device=dev, dtype=torch.int64)
                        check(lib.sn_linear_forward_maxpool(R, Ci, Co, N, ptr(a_in), ptr(coef_prev), ptr(W), ptr(b), ptr(z),
                                                            ptr(keys), ptr(pooled), ptr(argsel), ptr(zsel), int(tile_keys is not None), st),
                              "sn_linear_forward_maxpool")
                    zs.append(z)
                    idents.append(_ident(Co, pooled))
                    break
                z = torch.empty(R, Co, device=dev, dtype=torch.float32)
                check(lib.sn_linear_forward(R, Ci, Co, ptr(a_in), ptr(coef_prev), ptr(W), ptr(b), ptr(z), None, st),
                      "sn_linear_forward")
                zs.append(z)
                coef_prev = _ident(Co, z)
                idents.append(coef_prev)
                a_in = z
            else:
                check(lib.sn_pool_forward(B, N, C, ptr(zs[-1]), ptr(idents[-1]), ptr(pooled), ptr(argsel), ptr(zsel), st),
                      "sn_pool_forward")
        ctx.zlast_missing = zs[-1] is None
        ctx.zlast_missing_front = any(z is None for z in zs[:-1])
        zs = [pooled if z is None else z for z in zs]  # (placeholders in the saved list: never read -- see keep_z / need_grad)
        ctx.save_for_backward(x_bnc, pooled, argsel, zsel, *zs, *idents, *Ws)
        ctx.nl = len(Ws)
        return pooled

    @staticmethod
    def backward(ctx, g):
        nl = ctx.nl
        saved = ctx.saved_tensors
        x_bnc, pooled, argsel, zsel = saved[:4]
        zs, idents, Ws = saved[4:4 + nl], saved[4 + nl:4 + 2 * nl], saved[4 + 2 * nl:]
        B, N, _ = x_bnc.shape
        R = B * N
        dev = x_bnc.device
        g = g.contiguous().float()
        grads = [None] * (2 * nl)
        with torch.cuda.device(dev):
            st = _st(x_bnc)
            C = Ws[-1].shape[0]
            sparse = _sparse_last(ctx.needs_input_grad, nl, B, N, Ws)
            if ctx.zlast_missing and not sparse:
                raise RuntimeError("PointNetFeatures: the last layer's activations were not kept (test hooks changed between "
                                   "forward and backward?)")
            gsel = scratch = None
            if not sparse:
                gsel = torch.empty(B, C, device=dev, dtype=torch.float32)
                scratch = torch.empty(2 * C, device=dev, dtype=torch.float32)
                check(lib.sn_pool_backward(B, C, ptr(g), ptr(pooled), ptr(zsel), ptr(gsel), ptr(scratch), st), "sn_pool_backward")
            dy = None
            for i in range(nl - 1, -1, -1):
                W = Ws[i]
                Co, Ci = W.shape[0], W.shape[1]
                if sparse and i == nl - 1:
                    # dZ of this layer is one non-zero per cloud and channel: pooling backward + data gradient in one launch
                    dy = torch.empty(R, Ci, device=dev, dtype=torch.float32)
                    check(lib.sn_pool_dgrad_sparse(B, N, Ci, Co, ptr(g), ptr(pooled), ptr(argsel), ptr(W), ptr(zs[i - 1]),
                                                   ptr(idents[i - 1]), ptr(dy), st), "sn_pool_dgrad_sparse")
                    continue
                if (i == 3 and nl == 5 and FUSE_NARROW and ctx.needs_input_grad[0] and not any(ctx.needs_input_grad[1:9])
                        and Ws[0].shape[1] == 3 and lib.sn_pointnet_narrow_backward_supported(R, *[w.shape[0] for w in Ws[:4]])
                        and not ctx.zlast_missing_front):
                    # frozen conv1..conv4: their four data-gradient launches as one, straight to the gradient of the cloud
                    planes, ready = _weight_planes(Ws[3], Ws[2], Ws[1], tag="T")
                    dx = torch.empty(R, 3, device=dev, dtype=torch.float32)
                    check(lib.sn_pointnet_narrow_backward(R, ptr(dy), ptr(zs[0]), ptr(zs[1]), ptr(zs[2]), ptr(Ws[0]), ptr(Ws[1]),
                                                          ptr(Ws[2]), ptr(Ws[3]), ptr(planes), int(ready), ptr(dx), st),
                          "sn_pointnet_narrow_backward")
                    dy = dx
                    break
                mode = _DZ_POOL if i == nl - 1 else _DZ_PLAIN
                kcoef = None
                if mode == _DZ_POOL:  # dZ = 1 * dY_sparse + 0 * Z + 0
                    kcoef = _unit_rows(3, Co, x_bnc)
                aprev = zs[i - 1] if i > 0 else x_bnc.reshape(R, 3)
                cprev = idents[i - 1] if i > 0 else None
                gs, ag = (gsel, argsel) if mode == _DZ_POOL else (None, None)
                if ctx.needs_input_grad[1 + 2 * i] or ctx.needs_input_grad[2 + 2 * i]:
                    ns = lib.sn_linear_wgrad_splits(R, Ci, Co, 1)
                    part = torch.empty(ns * Co * (Ci + 1), device=dev, dtype=torch.float32)
                    dW = torch.empty_like(W)
                    db = torch.empty(Co, device=dev, dtype=torch.float32)
                    check(lib.sn_linear_wgrad(R, Ci, Co, mode, ptr(dy), ptr(zs[i]), ptr(kcoef), ptr(gs), ptr(ag), N, ptr(aprev),
                                              ptr(cprev), ptr(part), ptr(dW), ptr(db), st), "sn_linear_wgrad")
                    grads[2 * i], grads[2 * i + 1] = dW, db
                if i > 0 or ctx.needs_input_grad[0]:
                    dyprev = torch.empty(R, Ci, device=dev, dtype=torch.float32)
                    nblk = lib.sn_linear_stats_blocks(R)
                    stats = torch.empty(nblk * 2 * Ci, device=dev, dtype=torch.float32) if i > 0 else None
                    check(lib.sn_linear_dgrad(R, Ci, Co, mode, ptr(dy), ptr(zs[i]), ptr(kcoef), ptr(gs), ptr(ag), N, ptr(W),
                                              ptr(aprev), ptr(cprev), ptr(dyprev), ptr(stats), st), "sn_linear_dgrad")
                    dy = dyprev
        gx = dy.reshape(B, N, 3) if ctx.needs_input_grad[0] else None
        return (gx,) + tuple(grads)


class PointNetFeatures(nn.Module):
    def __init__(self, bottleneck_size=1024, input_shape="bcn"):
        super().__init__()
        if input_shape not in ["bcn", "bnc"]:
            raise ValueError("allowed shape are 'bcn' (batch * channels * num_in_points), 'bnc' ")
        self.input_shape = input_shape
        self.conv1 = torch.nn.Conv1d(3, 64, 1)
        self.conv2 = torch.nn.Conv1d(64, 64, 1)
        self.conv3 = torch.nn.Conv1d(64, 64, 1)
        self.conv4 = torch.nn.Conv1d(64, 128, 1)
        self.conv5 = torch.nn.Conv1d(128, bottleneck_size, 1)

    def forward(self, x):
        if self.input_shape == "bcn":
            x = x.permute(0, 2, 1)
        if x.shape[2] != 3:
            raise RuntimeError("shape of x must be of [Batch x 3 x NumInPoints]")
        if not x.is_cuda:
            raise RuntimeError("samplenet_amd.task_features runs on the GPU only; no CPU fallback exists")
        wb = []
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            wb += [conv.weight.reshape(conv.weight.shape[0], conv.weight.shape[1]), conv.bias]
        return _FeaturesFunction.apply(x.contiguous().float(), *wb)


def _skinny_scratch(nbytes, ntiles, like):
    """Slice partials + per-tile arrival counters of sn_skinny_linear (the counters start and are left at zero); one pair per
    device and stream serves every layer (launches on a stream run one after the other); built for the call only while a
    stream capture is under way (see _unit_rows)."""
    key = ("skinny", like.device, torch.cuda.current_stream(like.device).cuda_stream)
    t = _CONST.get(key)
    if t is None or t[0].numel() * 4 < nbytes or t[1].numel() < ntiles:
        t = (torch.empty(max(1, nbytes // 4), device=like.device, dtype=torch.float32),
             torch.zeros(max(64, ntiles), device=like.device, dtype=torch.int32))
        if not torch.cuda.is_current_stream_capturing():
            _CONST[key] = t
    return t


def _trunk_scratch(R, Ws, like):
    """One (partials, counters) pair large enough for every layer of a trunk in either direction: the launches of a pass run one after
    the other on the stream and leave the counters at zero, so they share it -- under a stream capture that is ONE allocation and
    zero-fill per pass instead of one per layer (4.7 us each in a replay)."""
    nbytes = ntiles = 0
    for W in Ws:
        Co, Ci = W.shape
        nbytes = max(nbytes, lib.sn_skinny_linear_scratch_bytes(R, Ci, Co), lib.sn_skinny_linear_scratch_bytes(R, Co, Ci))
        ntiles = max(ntiles, (Co + 31) // 32, (Ci + 31) // 32)
    return _skinny_scratch(nbytes, ntiles, like)


def _skinny(x, gate, W, transposed, bias, relu, x2=None, split_out=None, scratch=None, st=None):
    """out (R, N) = act((x . [gate > 0]) W^T + bias) with W (N, K), or x W with W (K, N) when transposed.  x2: the input is
    [x | x2] (two tensors, never concatenated).  split_out = (n0, want0, want1): the output leaves as two tensors (R, n0) and
    (R, N - n0), each only if wanted -> (out0 | None, out1 | None)."""
    R = x.shape[0]
    K = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    N = W.shape[1] if transposed else W.shape[0]
    part, counters = scratch if scratch is not None else _skinny_scratch(lib.sn_skinny_linear_scratch_bytes(R, K, N), (N + 31) // 32, x)
    if split_out is None:
        out, out2, nsplit = torch.empty(R, N, device=x.device, dtype=torch.float32), None, 0
    else:
        nsplit, want0, want1 = split_out
        out = torch.empty(R, nsplit, device=x.device, dtype=torch.float32) if want0 else None
        out2 = torch.empty(R, N - nsplit, device=x.device, dtype=torch.float32) if want1 else None
    check(lib.sn_skinny_linear2(R, K, N, ptr(x), ptr(x2), x.shape[1] if x2 is not None else 0, ptr(gate), ptr(W), int(transposed),
                                ptr(bias), int(relu), ptr(out), ptr(out2), nsplit, ptr(part), ptr(counters), st if st is not None else _st(x)),
          "sn_skinny_linear2")
    return out if split_out is None else (out, out2)


def _skinny_wgrad(x, x2, dy, gate, W, st):
    """dW (N, K), db (N) of a layer out = act([x | x2] W^T + b) from dy (R, N) and the layer's own output `gate` (ReLU mask)."""
    R, N = dy.shape
    K = W.shape[1]
    dW = torch.empty_like(W)
    db = torch.empty(N, device=W.device, dtype=torch.float32)
    check(lib.sn_skinny_wgrad(R, K, N, ptr(x), ptr(x2), x.shape[1] if x2 is not None else 0, ptr(dy), ptr(gate), ptr(dW), ptr(db), st),
          "sn_skinny_wgrad")
    return dW, db


class _TrunkFunction(torch.autograd.Function):
    """PCRNet's FC trunk on at most 128 rows: [f0 | f1] -> fc1 .. fc5 (ReLU) -> fc6, six sn_skinny_linear launches forward and six
    for the data gradient (registration/models/pcrnet.py:56-77 as rocBLAS GEMMs + ReLU / mask kernels: 22 launches); weights that
    want a gradient get it from six sn_skinny_wgrad launches (frozen weights -- the sampler's training step -- skip them).
    The two clouds' feature vectors are read where they lie and each receives its own gradient tensor (no cat / slice copies)."""

    @staticmethod
    def forward(ctx, f0, f1, *wb):
        Ws, bs = wb[0::2], wb[1::2]
        f0, f1 = f0.contiguous().float(), f1.contiguous().float()
        acts = []
        with torch.cuda.device(f0.device):
            sc = _trunk_scratch(f0.shape[0], Ws, f0)
            st = _st(f0)  # (one stream lookup per pass: torch.cuda.current_stream costs ~5 us of host time)
            x = _skinny(f0, None, Ws[0], False, bs[0], True, x2=f1, scratch=sc, st=st)
            acts.append(x)
            for i in range(1, len(Ws)):
                x = _skinny(x, None, Ws[i], False, bs[i], i < len(Ws) - 1, scratch=sc, st=st)
                acts.append(x)
        ctx.save_for_backward(*acts[:-1], *Ws)
        ctx.inputs = (f0, f1) if any(ctx.needs_input_grad[2:]) else (None, None)  # (the first layer's weight gradient reads them)
        ctx.nl = len(Ws)
        ctx.n0 = f0.shape[1]
        ctx.sc = sc  # (the backward's launches reuse the pair: the counters are back at zero)
        return x

    @staticmethod
    def backward(ctx, g):
        nl = ctx.nl
        acts, Ws = ctx.saved_tensors[:nl - 1], ctx.saved_tensors[nl - 1:]
        g = g.contiguous().float()
        with torch.cuda.device(g.device):
            sc = ctx.sc if ctx.sc[0].device == g.device else _trunk_scratch(g.shape[0], Ws, g)
            st = _st(g)
            wgrads = [None] * (2 * nl)
            f0, f1 = ctx.inputs
            for i in range(nl - 1, 0, -1):  # dX = (dY . [y_i > 0]) W_i; the last layer has no ReLU
                gate = acts[i] if i < nl - 1 else None
                if ctx.needs_input_grad[2 + 2 * i] or ctx.needs_input_grad[3 + 2 * i]:  # trainable trunk: dW_i, db_i
                    wgrads[2 * i], wgrads[2 * i + 1] = _skinny_wgrad(acts[i - 1], None, g, gate, Ws[i], st)
                g = _skinny(g, gate, Ws[i], True, None, False, scratch=sc, st=st)
            if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
                wgrads[0], wgrads[1] = _skinny_wgrad(f0, f1, g, acts[0], Ws[0], st)
            g0, g1 = _skinny(g, acts[0], Ws[0], True, None, False, split_out=(ctx.n0, ctx.needs_input_grad[0], ctx.needs_input_grad[1]),
                             scratch=sc, st=st)
        return (g0, g1) + tuple(wgrads)


TRUNK_ROWS = 128  # rows of one trunk pass (sn_skinny_linear multiplies the weight fragments with up to four 32-row tiles)


def _trunk(f0, f1, wb):
    """The FC trunk on any number of rows: blocks of TRUNK_ROWS rows, one _TrunkFunction pass each (the weights are streamed once
    per block: 15.5 MB against ~1 MFLOP per row -- above a few hundred rows a tile GEMM would win, but nothing on the path gets
    there: registration/main.py evaluates the network on batches of 32); per-row arithmetic does not depend on the blocking, so
    the outputs are bit-identical to separate evaluations of the blocks; weight gradients of a trainable trunk are the blocks'
    gradients added up by autograd in block order."""
    R = f0.shape[0]
    if R <= TRUNK_ROWS:
        return _TrunkFunction.apply(f0, f1, *wb)
    return torch.cat([_TrunkFunction.apply(f0[a:a + TRUNK_ROWS], f1[a:a + TRUNK_ROWS], *wb) for a in range(0, R, TRUNK_ROWS)], dim=0)


class _HeadFunction(torch.autograd.Function):
    """y (B,7) -> twist (B,7) = [normalize(y[:, 0:4]) | y[:, 4:7]], quat (B,4) = the normalised quaternion as its own contiguous
    tensor (for the rotation: no slice / copy launches, no zero-padded slice gradient), qnorm = mean_b (||y[:, 0:4]||^2 - 1)^2
    -- sn_pcrnet_head_*."""

    @staticmethod
    def forward(ctx, y):
        y = y.contiguous().float()
        B = y.shape[0]
        twist = torch.empty_like(y)
        quat = torch.empty(B, 4, device=y.device, dtype=torch.float32)
        qnorm = torch.empty((), device=y.device, dtype=torch.float32)
        with torch.cuda.device(y.device):
            check(lib.sn_pcrnet_head_forward(B, ptr(y), ptr(twist), ptr(quat), ptr(qnorm), _st(y)), "sn_pcrnet_head_forward")
        ctx.save_for_backward(y)
        ctx.set_materialize_grads(False)  # (an unused output's gradient arrives as None, not as a zero-filled tensor: a fill launch each)
        return twist, quat, qnorm

    @staticmethod
    def backward(ctx, g_twist, g_quat, g_qnorm):
        (y,) = ctx.saved_tensors
        gy = torch.empty_like(y)
        gt = g_twist.contiguous().float() if g_twist is not None else None
        gqt = g_quat.contiguous().float() if g_quat is not None else None
        gq = g_qnorm.contiguous().float() if g_qnorm is not None else None
        with torch.cuda.device(y.device):
            check(lib.sn_pcrnet_head_backward(y.shape[0], ptr(y), ptr(gt), ptr(gqt), ptr(gq), ptr(gy), _st(y)),
                  "sn_pcrnet_head_backward")
        return gy


class _HeadRotFunction(torch.autograd.Function):
    """_HeadFunction + _QrotCloudFunction as ONE launch each way: y (B,7), v (B,N,3) -> twist (B,7), quat (B,4), qnorm (),
    rotated (B,N,3) = qrot(quat, v)  (sn_pcrnet_head_rot_*; bit-identical to the two launches)."""

    @staticmethod
    def forward(ctx, y, v):
        y, x = y.contiguous().float(), v.contiguous().float()
        B, N, _ = x.shape
        twist = torch.empty_like(y)
        quat = torch.empty(B, 4, device=y.device, dtype=torch.float32)
        qnorm = torch.empty((), device=y.device, dtype=torch.float32)
        out = torch.empty_like(x)
        with torch.cuda.device(y.device):
            check(lib.sn_pcrnet_head_rot_forward(B, N, ptr(y), ptr(x), ptr(twist), ptr(quat), ptr(qnorm), ptr(out), _st(y)),
                  "sn_pcrnet_head_rot_forward")
        ctx.save_for_backward(y, quat, x)
        ctx.set_materialize_grads(False)
        return twist, quat, qnorm, out

    @staticmethod
    def backward(ctx, g_twist, g_quat, g_qnorm, g_out):
        y, quat, x = ctx.saved_tensors
        B, N, _ = x.shape
        gy = torch.empty_like(y)
        gt = g_twist.contiguous().float() if g_twist is not None else None
        gqt = g_quat.contiguous().float() if g_quat is not None else None
        gq = g_qnorm.contiguous().float() if g_qnorm is not None else None
        go = g_out.contiguous().float() if g_out is not None else None
        gv = torch.empty_like(x) if (ctx.needs_input_grad[1] and go is not None) else None
        with torch.cuda.device(y.device):
            check(lib.sn_pcrnet_head_rot_backward(B, N, ptr(y), ptr(quat), ptr(x), ptr(go), ptr(gt), ptr(gqt), ptr(gq), ptr(gv), ptr(gy),
                                                  _st(y)), "sn_pcrnet_head_rot_backward")
        if ctx.needs_input_grad[1] and gv is None:
            gv = torch.zeros_like(x)
        return gy, gv


class _HeadRotGroupedFunction(torch.autograd.Function):
    """_HeadRotFunction for E evaluations against the same `group` template clouds in one launch each way: y (E group, 7),
    v (group, N, 3) [no gradient] -> twist (E group, 7), quat (E group, 4), qnorm (E,), rotated (E group, N, 3); every row equals its
    evaluation's own _HeadRotFunction call (sn_pcrnet_head_rot_*_grouped)."""

    @staticmethod
    def forward(ctx, y, v, group):
        y, x = y.contiguous().float(), v.contiguous().float()
        R, N = y.shape[0], x.shape[1]
        twist = torch.empty_like(y)
        quat = torch.empty(R, 4, device=y.device, dtype=torch.float32)
        qnorm = torch.empty(R // group, device=y.device, dtype=torch.float32)
        out = torch.empty(R, N, 3, device=y.device, dtype=torch.float32)
        with torch.cuda.device(y.device):
            check(lib.sn_pcrnet_head_rot_forward_grouped(R, N, int(group), ptr(y), ptr(x), ptr(twist), ptr(quat), ptr(qnorm), ptr(out),
                                                         _st(y)), "sn_pcrnet_head_rot_forward_grouped")
        ctx.save_for_backward(y, quat, x)
        ctx.group = int(group)
        ctx.set_materialize_grads(False)
        return twist, quat, qnorm, out

    @staticmethod
    def backward(ctx, g_twist, g_quat, g_qnorm, g_out):
        y, quat, x = ctx.saved_tensors
        R, N = y.shape[0], x.shape[1]
        gy = torch.empty_like(y)
        gt = g_twist.contiguous().float() if g_twist is not None else None
        gqt = g_quat.contiguous().float() if g_quat is not None else None
        gq = g_qnorm.contiguous().float() if g_qnorm is not None else None
        go = g_out.contiguous().float() if g_out is not None else None
        with torch.cuda.device(y.device):
            check(lib.sn_pcrnet_head_rot_backward_grouped(R, N, ctx.group, ptr(y), ptr(quat), ptr(x), ptr(go), ptr(gt), ptr(gqt), ptr(gq),
                                                          ptr(gy), _st(y)), "sn_pcrnet_head_rot_backward_grouped")
        return gy, None, None


class PCRNet(nn.Module):
    """Drop-in for `registration/models/pcrnet.py:44-82` (same constructor, attribute and parameter names -> state_dict
    compatible, same `forward(x0, x1) -> (twist (B,7), pre_normalized_quat (B,4))`): the two feature extractions run on the
    HIP MLP kernels (`PointNetFeatures` above, the 99 % of the network's arithmetic); the six-layer FC trunk on B rows runs on
    `sn_skinny_linear` (forward and data gradient) in row blocks of up to 128 -- frozen as in the sampler's training step, or
    trainable (main.py --train-pcrnet: weight / bias gradients on `sn_skinny_wgrad`); no library GEMM (rocBLAS) on the GPU path at
    any batch; the output head is one launch each way (`sn_pcrnet_head_*`)."""

    # registration/main.py:296 hangs the (trainable) sampler on the task network as `model.sampler`; PCRNet.forward never calls
    # it, so the captured calls of the frozen network (graphed.py) do not own its parameters
    _graphed_exclude = ("sampler",)

    def __init__(self, bottleneck_size=1024, input_shape="bcn"):
        super().__init__()
        if input_shape not in ["bcn", "bnc"]:
            raise ValueError("allowed shape are 'bcn' (batch * channels * num_in_points), 'bnc' ")
        self.input_shape = input_shape
        self.feat = PointNetFeatures(bottleneck_size, input_shape)
        self.fc1 = nn.Linear(bottleneck_size * 2, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        self.fc3 = nn.Linear(1024, 512)
        self.fc4 = nn.Linear(512, 512)
        self.fc5 = nn.Linear(512, 256)
        self.fc6 = nn.Linear(256, 7)

    def static_weights(self, flag=True):
        """Declare this network's weights CONSTANT for graphs the caller captures around it (engine.SamplerTrainStep(task_loss=...),
        torch.cuda.graph): the bf16 planes of the frozen weights are then split once, outside the graphs, instead of by four launches
        of every replay (registration/main.py:272-277 freezes the task network for the sampler's training).  A promise: after
        changing a weight in place (load_state_dict, an optimizer step) such graphs must be captured again -- the frozen network's
        own captured calls (graphed.py) need no promise, they check the parameters' version counters themselves.  -> self."""
        for p in self.parameters():
            b = p._base if p._base is not None else p
            if flag:
                _STATIC[id(b)] = weakref.ref(b, lambda _r, k=id(b): _STATIC.pop(k, None))
            else:
                _STATIC.pop(id(b), None)
        return self

    def __getstate__(self):  # (copy.deepcopy / pickling: captured graphs stay behind)
        return {k: v for k, v in self.__dict__.items() if k != "_sn_graphed"}

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_sn_graphed", None)  # .to() / .cuda(): the graphs' addresses are void
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x0, x1):
        # a frozen network under a sampler's training step (main.py:557-563): replayed from two captured graphs once the
        # configuration has been seen (graphed.py); None: op by op
        from . import graphed

        out = graphed.call(self, "forward", self._forward, (x0, x1))
        return out if out is not None else self._forward(x0, x1)

    def _forward(self, x0, x1):
        twist, pre_normalized_quat = self.forward_with_qnorm(x0, x1)[:2]
        return twist, pre_normalized_quat

    def template_features(self, x0):
        """self.feat(x0) for a caller that evaluates the network several times against the SAME template cloud within a step (the
        progressive sampler's prefixes): pass the result as `feat0` and the template's extractor pass -- two thirds of the
        network's arithmetic -- runs once."""
        return self.feat(x0)

    def forward_multi(self, x0, x1_list, feat0=None, rotate=None):
        """forward_with_qnorm for several source clouds against ONE template (the progressive sampler's prefixes) with the FC trunk
        run once on all of them: the trunk is a stream of 15.5 MB of weights per pass whatever the number of rows, so E evaluations of
        B clouds cost one pass on E B rows (at most 128) instead of E passes.  -> list of (twist, pre_normalized_quat, qnorm, quat
        [, rotate rotated by quat: see forward_with_qnorm])."""
        E = len(x1_list)
        f0 = self.template_features(x0) if feat0 is None else feat0
        B = f0.shape[0]
        fcs = (self.fc1, self.fc2, self.fc3, self.fc4, self.fc5, self.fc6)
        if not (E * B <= 128 and f0.shape[1] % 8 == 0 and E > 1):
            return [self.forward_with_qnorm(x0, x1, feat0=f0, rotate=rotate) for x1 in x1_list]
        f1 = self._feat_multi(x1_list)
        wb = []
        for fc in fcs:
            wb += [fc.weight, fc.bias]
        y = _TrunkFunction.apply(f0.repeat(E, 1), f1, *wb)  # (E B, 7)
        out = []
        for ye in y.view(E, B, -1).unbind(0):  # (one stack in the backward instead of zeros + copy + add per evaluation)
            if rotate is not None:
                twist, quat, qnorm, rotated = _HeadRotFunction.apply(ye, rotate)
                out.append((twist, ye[:, 0:4], qnorm, quat, rotated))
            else:
                twist, quat, qnorm = _HeadFunction.apply(ye)
                out.append((twist, ye[:, 0:4], qnorm, quat))
        return out

    def _one_batch_ok(self, x0, x1_list, feat0=None):
        """Several evaluations against one template as ONE batch through extractor, trunk, head + rotation and Chamfer term
        (_pcrnet_chamfer_loss_multi): 'bnc' clouds on the GPU, at most 16 of them and 128 trunk rows, the padded batch below ~2.5 x
        the clouds' own points, at most 2048 points a side (the grouped Chamfer backward keeps a cloud in registers)."""
        E = len(x1_list)
        if not (1 < E <= 16 and self.input_shape == "bnc" and x0.is_cuda and type(self.feat) is PointNetFeatures):
            return False
        if x0.requires_grad and torch.is_grad_enabled():  # (the grouped head + rotation hands no gradient to the shared template)
            return False
        sizes = [x.shape[1] for x in x1_list]
        B = x0.shape[0]
        K2 = self.fc1.in_features // 2
        return (E * B <= 128 and K2 % 8 == 0 and max(sizes) * E <= 2.5 * sum(sizes) and max(sizes) <= 2048 and x0.shape[1] <= 2048
                and all(x.dim() == 3 and x.shape[0] == B and x.shape[2] == 3 for x in x1_list)
                and not any(p.requires_grad for p in self.parameters()))

    def forward_multi_one_batch(self, x0, x1_list, feat0=None):
        """forward_multi with everything batched: -> (twist (E B, 7), y (E B, 7), qnorm (E,), quat (E B, 4), rotated template
        (E B, N, 3), the clouds padded to one size (E B, P, 3)).  Row e B + b = evaluation e, cloud b."""
        from .ops import cyclic_pad_cat

        E = len(x1_list)
        f0 = self.template_features(x0) if feat0 is None else feat0
        B = f0.shape[0]
        xpad = cyclic_pad_cat(x1_list)
        f1 = self.feat(xpad)
        wb = []
        for fc in (self.fc1, self.fc2, self.fc3, self.fc4, self.fc5, self.fc6):
            wb += [fc.weight, fc.bias]
        y = _TrunkFunction.apply(f0.repeat(E, 1), f1, *wb)
        twist, quat, qnorm, rotated = _HeadRotGroupedFunction.apply(y, x0, B)
        return twist, y, qnorm, quat, rotated, xpad

    def _feat_multi(self, x1_list):
        """self.feat of several source clouds, concatenated over the batch.  Clouds of DIFFERENT sizes (the progressive sampler's
        prefixes) go through the extractor as ONE batch: every cloud repeated cyclically up to the largest size (ops.cyclic_pad_cat) --
        the extractor has no BatchNorm and reduces over the points with a maximum only, so the copies change nothing (bit for bit) --
        instead of one latency-bound pass per cloud.  Taken while the padded batch stays below ~2.5 x the clouds' own points."""
        sizes = [x.shape[1] if self.input_shape == "bnc" else x.shape[2] for x in x1_list]
        P = max(sizes)
        if (len(x1_list) > 1 and len(set(sizes)) > 1 and len(x1_list) <= 16 and self.input_shape == "bnc" and x1_list[0].is_cuda
                and P * len(sizes) <= 2.5 * sum(sizes) and type(self.feat) is PointNetFeatures):
            from .ops import cyclic_pad_cat

            return self.feat(cyclic_pad_cat(x1_list))  # (E B, K)
        return torch.cat([self.feat(x1) for x1 in x1_list], dim=0)

    def forward_with_qnorm(self, x0, x1, feat0=None, rotate=None):
        """forward() plus the QuaterNet regulariser mean((||pre_normalized_quat||^2 - 1)^2) of registration/main.py:565 and the
        normalised quaternion as a contiguous (B,4) tensor, both of which the output head's kernel produces on the side:
        (twist, pre_normalized_quat, qnorm, quat).  feat0: template_features(x0), computed by the caller (x0 is then unused).
        rotate: a (B,N,3) cloud (bnc) -- the head's launch also rotates it by the estimated quaternion (main.py:569-571
        est_transform.rotate(p0)); the rotated cloud is returned as a fifth element."""
        f0, f1 = (self.feat(x0) if feat0 is None else feat0), self.feat(x1)
        fcs = (self.fc1, self.fc2, self.fc3, self.fc4, self.fc5, self.fc6)
        wb = []
        for fc in fcs:
            wb += [fc.weight, fc.bias]
        if f0.shape[1] % 8:
            # the kernels read the first layer's input as two parts that meet at a multiple of 8 columns: a bottleneck width
            # that is not one is re-cut there (copies of B x 2 C values; the layers themselves stay on sn_skinny_linear)
            cat = torch.cat([f0, f1], dim=1)
            k8 = max(8, (cat.shape[1] // 2) // 8 * 8)
            f0, f1 = cat[:, :k8].contiguous(), cat[:, k8:].contiguous()
        y = _trunk(f0, f1, wb)  # (B, 7)
        pre_normalized_quat = y[:, 0:4]
        if rotate is not None:
            twist, quat, qnorm, rotated = _HeadRotFunction.apply(y, rotate)
            return twist, pre_normalized_quat, qnorm, quat, rotated
        twist, quat, qnorm = _HeadFunction.apply(y)
        return twist, pre_normalized_quat, qnorm, quat


def pcrnet_chamfer_loss_multi(model, p0, p1_list, template_features=None):
    """pcrnet_chamfer_loss for several source clouds against one template, the network evaluated by model.forward_multi (one
    trunk pass for all of them).  -> list of (chamfer_loss, qnorm_loss, twist)."""
    if not hasattr(model, "forward_multi"):
        return [pcrnet_chamfer_loss(model, p0, p1, template_features) for p1 in p1_list]
    if template_features is None and isinstance(model, nn.Module) and len(p1_list) > 1:
        # a frozen network: template extractor + the E evaluations + their losses replay two captured graphs (graphed.py)
        from . import graphed

        E = len(p1_list)

        def flat(a, *ps):
            res = _pcrnet_chamfer_loss_multi(model, a, list(ps), None)
            return tuple(t for triple in res for t in triple)

        got = graphed.call(model, "pcrnet_chamfer_loss_multi%d" % E, flat, (p0,) + tuple(p1_list))
        if got is not None:
            return [tuple(got[3 * e:3 * e + 3]) for e in range(E)]
    return _pcrnet_chamfer_loss_multi(model, p0, p1_list, template_features)


def _pcrnet_chamfer_loss_multi(model, p0, p1_list, template_features):
    from .ops import chamfer_mean_loss, chamfer_mean_loss_grouped

    if hasattr(model, "_one_batch_ok") and model._one_batch_ok(p0, p1_list, template_features):
        # every evaluation in ONE batch: extractor pass, trunk pass, head + rotation launch, Chamfer scan and reductions -- the clouds
        # padded to one size by cyclic repetition, the copies left out of the loss; the same numbers as evaluation by evaluation
        E, B = len(p1_list), p0.shape[0]
        twist, _y, qnorm, _quat, rotated, xpad = model.forward_multi_one_batch(p0, p1_list, feat0=template_features)
        losses = chamfer_mean_loss_grouped(xpad, rotated, B, [p.shape[1] for p in p1_list])
        return list(zip(losses.unbind(0), qnorm.unbind(0), twist.view(E, B, -1).unbind(0)))
    out = []
    for p1, (twist, _pre, qnorm, _quat, p1_est) in zip(p1_list, model.forward_multi(p0, p1_list, feat0=template_features, rotate=p0)):
        out.append((chamfer_mean_loss(p1.contiguous(), p1_est.contiguous()), qnorm, twist))
    return out


def qrot(q, v):
    """Rotate v (*,3) by the quaternion q (*,4), (w, x, y, z) order -- registration/src/quaternion.py:35-53."""
    qvec = q[..., 1:]
    uv = torch.cross(qvec, v, dim=-1)
    uuv = torch.cross(qvec, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


class _QrotCloudFunction(torch.autograd.Function):
    """out (B,N,3) = qrot(quat (B,4) expanded over the points, v (B,N,3)) -- sn_qrot_forward / sn_qrot_backward."""

    @staticmethod
    def forward(ctx, quat, v):
        q, x = quat.contiguous().float(), v.contiguous().float()
        B, N, _ = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.sn_qrot_forward(B, N, ptr(q), ptr(x), ptr(out), _st(x)), "sn_qrot_forward")
        ctx.save_for_backward(q, x)
        return out

    @staticmethod
    def backward(ctx, g):
        q, x = ctx.saved_tensors
        B, N, _ = x.shape
        g = g.contiguous().float()
        gq = torch.empty_like(q) if ctx.needs_input_grad[0] else None
        gv = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        if gq is None and gv is None:
            return None, None
        with torch.cuda.device(x.device):
            check(lib.sn_qrot_backward(B, N, ptr(q), ptr(x), ptr(g), ptr(gq), ptr(gv), _st(x)), "sn_qrot_backward")
        return gq, gv


def qrot_cloud(quat, v):
    """Rotate every cloud v[b] (N,3) by its quaternion quat[b] (w, x, y, z): qrot(quat.unsqueeze(1).expand(-1, N, -1), v) in one
    launch (and one for the backward) instead of the elementwise chain of `qrot`."""
    if not v.is_cuda:
        raise RuntimeError("samplenet_amd.task_features runs on the GPU only; no CPU fallback exists")
    return _QrotCloudFunction.apply(quat, v)


def pcrnet_chamfer_loss(model, p0, p1, template_features=None):
    """The Chamfer term of the registration task loss (`registration/main.py:557-577`, `--loss-type 1`):
    twist = model(p0, p1); p1_est = rotate(p0) by the estimated quaternion (QuaternionTransform.rotate,
    qdataset.py:97-119: rotation only); loss = mean d(p1 -> p1_est) + mean d(p1_est -> p1) on the HIP Chamfer kernels.
    p0 template / p1 source, (B,N,3).  Returns (chamfer_loss, qnorm_loss, twist).  The rotation-matrix error terms of
    `--loss-type 0` go through kornia in the reference (not installed here) and stay with the caller.
    template_features: model.template_features(p0), when several evaluations of a step share the template."""
    if template_features is None and isinstance(model, nn.Module):
        from . import graphed

        out = graphed.call(model, "pcrnet_chamfer_loss", lambda a, b: _pcrnet_chamfer_loss(model, a, b, None), (p0, p1))
        if out is not None:  # (frozen network: the whole task term replays two captured graphs, graphed.py)
            return out
    return _pcrnet_chamfer_loss(model, p0, p1, template_features)


def _pcrnet_chamfer_loss(model, p0, p1, template_features):
    from .ops import chamfer_mean_loss

    if hasattr(model, "forward_with_qnorm"):
        # (the output head's launch rotates the template as well: one launch each way instead of two)
        twist, _pre, qnorm_loss, _quat, p1_est = model.forward_with_qnorm(p0, p1, feat0=template_features, rotate=p0)
    else:
        twist, pre_normalized_quat = model(p0, p1)
        qnorm_loss = torch.mean((torch.sum(pre_normalized_quat ** 2, dim=1) - 1) ** 2)
        p1_est = qrot_cloud(twist[:, 0:4], p0)  # = qrot(twist[:, 0:4] expanded over the points, p0)
    # mean(d(p1 -> p1_est)) + mean(d(p1_est -> p1)): scan + one fused reduction, implicit-gradient backward
    return chamfer_mean_loss(p1.contiguous(), p1_est.contiguous()), qnorm_loss, twist
