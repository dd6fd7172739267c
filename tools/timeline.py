#!/usr/bin/env python3
"""Phase timeline of the MLP GEMM kernels on the GPU (debug build: tools/build_timeline_lib.sh).
Prints, per kernel shape, the distribution over workgroups of: start offset from the first workgroup,
t(load landed in LDS), t(MFMA loop done), t(stores issued), t(stores drained), t(end) -- all in microseconds."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("TL_LIB") or os.path.join(ROOT, "tools", "_ab", "libsamplenet_hip_tl.so"))
vp, i = ctypes.c_void_p, ctypes.c_int


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def dump_fused(name, nblocks):
    torch.cuda.synchronize()
    host = np.zeros((nblocks, 16), dtype=np.uint64)
    assert lib.sn_debug_timeline(host.ctypes.data_as(vp), nblocks, 0) == 0
    t = host.astype(np.float64) / 100.0
    valid = host[:, 0] > 0
    t0 = t[valid, 0].min()
    names = ["start", "it1 MFMAs done", "it1 epilogue done", "it1 next tile staged", "it1 barrier passed", "-", "loop done", "end (drained)"]
    print("== %s: %d workgroups, span %.2f us" % (name, valid.sum(), t[valid, 7].max() - t0))
    hw = np.zeros((nblocks, 8), dtype=np.uint32)
    assert lib.sn_debug_timeline(hw.ctypes.data_as(vp), nblocks, 2) == 0
    simd = (hw >> 4) & 3
    waveid = hw & 15
    cu = (hw >> 8) & 15
    print("  wave -> SIMD of the first workgroups:", [list(map(int, simd[i])) for i in range(4)], " same CU:", bool((cu[:, :1] == cu).all()))
    for w, label in ((0, "dgrad wave 0"), (8, "wgrad wave 4")):
        print("  [%s]" % label)
        prev = None
        order = [0, 5, 1, 2, 3, 4, 6, 7]
        nm2 = dict(zip(range(8), names)); nm2.update({5: "it1 stores/loads issued", 1: "it1 MFMAs done", 2: "it1 epilogue done", 3: "it1 next tile staged", 4: "it1 barrier passed"})
        for sidx in order:
            nm = nm2[sidx]
            col = t[valid, w + sidx]
            if not (host[valid, w + sidx] > 0).all():
                continue
            d = (col - prev) if prev is not None else (col - t0)
            print("     %-26s +%.2f (p10 %.2f p90 %.2f)   abs med %.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(col - t0)))
            prev = col


def dump(name, nblocks):
    torch.cuda.synchronize()
    host = np.zeros((nblocks, 16), dtype=np.uint64)
    rc = lib.sn_debug_timeline(host.ctypes.data_as(vp), nblocks, 0)
    assert rc == 0
    t = host[:, :6].astype(np.float64) / 100.0  # 100 MHz -> us
    kind = (host[:, 7] >> np.uint64(48)).astype(int)
    xcc = ((host[:, 7] >> np.uint64(32)) & np.uint64(0xF)).astype(int)
    hw = (host[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    valid = host[:, 0] > 0
    t0 = t[valid, 0].min()
    print("== %s: %d workgroups, span %.2f us (first start -> last end)" % (name, valid.sum(), t[valid, 5].max() - t0))
    for k in sorted(set(kind[valid])):
        m = valid & (kind == k)
        label = {0: "fwd", 1: "wgrad", 2: "dgrad", 3: "fused bwd"}[k]
        st = t[m, 0] - t0
        print("  [%s] n=%d start: min %.2f med %.2f p90 %.2f max %.2f" % (label, m.sum(), st.min(), np.median(st), np.percentile(st, 90), st.max()))
        names = ["loaded", "mfma_done", "stores_issued", "drained", "end"] if k != 3 else ["tile0 staged", "tile0 computed", "tile1 staged", "all tiles done", "end"]
        prev = t[m, 0]
        for s, nm in zip(range(1, 6), names):
            if not (host[m, s] > 0).all():
                continue
            d = t[m, s] - prev
            print("     +%-14s med %.2f  p10 %.2f p90 %.2f max %.2f   (abs med %.2f)" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), np.median(t[m, s] - t0)))
            prev = t[m, s]
    # distinct (xcc, se, cu) used
    key = set(zip(xcc[valid], se[valid], cu[valid]))
    print("  distinct (xcc,se,cu): %d" % len(key))


def fwd(R, Ci, Co, reps=3):
    dev = "cuda"
    a = torch.randn(R, Ci, device=dev)
    coef = torch.rand(4, Ci, device=dev) + 0.5
    W = torch.randn(Co, Ci, device=dev) * 0.1
    b = torch.randn(Co, device=dev)
    z = torch.empty(R, Co, device=dev)
    nblk = lib.sn_linear_stats_blocks(R)
    stats = torch.empty(nblk, 2, Co, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for r in range(reps):
        lib.sn_debug_timeline(None, 0, 1)
        rc = lib.sn_linear_forward(R, Ci, Co, P(a), P(coef), P(W), P(b), P(z), P(stats), vp(st))
        assert rc == 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.sn_linear_forward(R, Ci, Co, P(a), P(coef), P(W), P(b), P(z), P(stats), vp(st))
    e1.record()
    torch.cuda.synchronize()
    print("fwd %dx%d->%d: %.2f us/launch back-to-back" % (R, Ci, Co, e0.elapsed_time(e1) * 1e3 / 20))
    lib.sn_debug_timeline(None, 0, 1)
    lib.sn_linear_forward(R, Ci, Co, P(a), P(coef), P(W), P(b), P(z), P(stats), vp(st))
    dump("linear_fwd R=%d %d->%d" % (R, Ci, Co), ((R + 63) // 64) * ((Co + 63) // 64))


def small_fwd(R, Ci, Co):
    global FUSED
    dev = "cuda"
    a = torch.randn(R, Ci, device=dev)
    coef = torch.rand(4, Ci, device=dev) + 0.5
    W = torch.randn(Co, Ci, device=dev) * 0.1
    b = torch.randn(Co, device=dev)
    z = torch.empty(R, Co, device=dev)
    stats = torch.empty(1, 2, Co, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for r in range(3):
        assert lib.sn_linear_forward(R, Ci, Co, P(a), P(coef), P(W), P(b), P(z), P(stats), vp(st)) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.sn_linear_forward(R, Ci, Co, P(a), P(coef), P(W), P(b), P(z), P(stats), vp(st))
    e1.record()
    torch.cuda.synchronize()
    print("small fwd %dx%d->%d: %.2f us/launch back-to-back" % (R, Ci, Co, e0.elapsed_time(e1) * 1e3 / 20))
    # cold: touch a lot of other memory first
    junk = torch.empty(64 << 20, device=dev); junk.fill_(1.0); torch.cuda.synchronize()
    lib.sn_debug_timeline(None, 0, 1)
    lib.sn_linear_forward(R, Ci, Co, P(a), P(coef), P(W), P(b), P(z), P(stats), vp(st))
    torch.cuda.synchronize()
    nb = (Co + 31) // 32
    host = np.zeros((nb, 16), dtype=np.uint64)
    assert lib.sn_debug_timeline(host.ctypes.data_as(vp), nb, 0) == 0
    t = host.astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    for nm, sidx in (("start", 0), ("loads landed", 1), ("MFMAs done", 2), ("wave sum done", 3), ("end", 5)):
        print("     %-16s abs med %.2f  (max %.2f)" % (nm, np.median(t[:, sidx] - t0), (t[:, sidx] - t0).max()))


def bwd(R, Ci, Co, mode, B=32):
    dev = "cuda"
    npts = R // B
    dy = torch.randn(R, Co, device=dev) if mode != 2 else None
    z = torch.randn(R, Co, device=dev)
    kc = torch.randn(3, Co, device=dev)
    gsel = torch.randn(B, Co, device=dev) if mode == 2 else None
    argsel = torch.randint(0, npts, (B, Co), device=dev, dtype=torch.int32) if mode == 2 else None
    W = torch.randn(Co, Ci, device=dev) * 0.1
    zprev = torch.randn(R, Ci, device=dev)
    coefp = torch.rand(4, Ci, device=dev) + 0.5
    dyprev = torch.empty(R, Ci, device=dev)
    nblk = lib.sn_linear_stats_blocks(R)
    stats = torch.empty(nblk, 2, Ci, device=dev)
    ns = lib.sn_linear_wgrad_splits(R, Ci, Co, 0)
    part = torch.empty(ns * Co * Ci, device=dev)
    dW = torch.empty(Co, Ci, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.sn_linear_backward(R, Ci, Co, mode, P(dy), P(z), P(kc), P(gsel), P(argsel), npts, P(W), P(zprev), P(coefp),
                                    P(dyprev), P(stats), P(part), P(dW), vp(st))
        assert rc == 0, rc

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("bwd %dx%d->%d mode %d: %.2f us/call (linear_bwd + wgrad_reduce) back-to-back, splits %d" % (R, Ci, Co, mode, e0.elapsed_time(e1) * 1e3 / 20, ns))
    lib.sn_debug_timeline(None, 0, 1)
    run()
    nw = ((Co + 63) // 64) * ((Ci + 63) // 64) * ns
    nd = ((R + 63) // 64) * ((Ci + 63) // 64)
    (dump_fused if FUSED else dump)("linear_bwd R=%d %d->%d mode %d" % (R, Ci, Co, mode), 256 if FUSED else min(16384, nw + nd))


def stack_fwd(B=32, N=1024):
    """The conv stack's forward as the sampler runs it (sn_conv_stack_forward_bn, statistics-chain kernels).  The timeline
    buffer is reset before the call and every GEMM kernel overwrites it, so the dump shows the LAST layer of the stack:
    truncated stacks show the earlier ones."""
    dev = "cuda"
    lib.sn_conv_stack_acc_elems.restype = ctypes.c_longlong
    lib.sn_conv_stack_forward_bn.argtypes = [i, i, i] + [vp] * 20
    for nl, chans in ((5, [3, 64, 64, 64, 128, 128]), (4, [3, 64, 64, 64, 128]), (2, [3, 64, 64])):
        R = B * N
        x = torch.rand(B, N, 3, device=dev) - 0.5
        Ws = [torch.randn(chans[l + 1], chans[l], device=dev) * 0.2 for l in range(nl)]
        bs = [torch.randn(chans[l + 1], device=dev) * 0.1 for l in range(nl)]
        gs = [torch.rand(chans[l + 1], device=dev) + 0.5 for l in range(nl)]
        bes = [torch.randn(chans[l + 1], device=dev) * 0.1 for l in range(nl)]
        rms = [torch.zeros(chans[l + 1], device=dev) for l in range(nl)]
        rvs = [torch.ones(chans[l + 1], device=dev) for l in range(nl)]
        nbt = [torch.zeros((), device=dev, dtype=torch.int64) for l in range(nl)]
        zs = [torch.empty(R, chans[l + 1], device=dev) for l in range(nl)]
        cs = [torch.empty(4, chans[l + 1], device=dev) for l in range(nl)]
        acc = torch.zeros(lib.sn_conv_stack_acc_elems(nl), device=dev, dtype=torch.int64)
        Cn = chans[-1]
        pv = torch.empty(R // 64, 2, Cn, device=dev)
        pi = torch.empty(R // 64, 2, Cn, device=dev, dtype=torch.int32)
        pooled, zsel = torch.empty(B, Cn, device=dev), torch.empty(B, Cn, device=dev)
        argsel = torch.empty(B, Cn, device=dev, dtype=torch.int32)
        VP = vp * nl
        arr = lambda ts: VP(*[t.data_ptr() for t in ts])  # noqa: E731
        ch = (ctypes.c_int * (nl + 1))(*chans)
        eps = (ctypes.c_float * nl)(*([1e-5] * nl))
        mom = (ctypes.c_float * nl)(*([0.1] * nl))
        st = torch.cuda.current_stream().cuda_stream

        def run():
            rc = lib.sn_conv_stack_forward_bn(B, N, nl, ch, P(x), arr(Ws), arr(bs), arr(gs), arr(bes), arr(rms), arr(rvs), arr(nbt), eps, mom,
                                              arr(zs), arr(cs), P(acc), P(pv), P(pi), P(pooled), P(argsel), P(zsel), vp(st))
            assert rc == 0, rc

        for _ in range(3):
            run()
        lib.sn_debug_timeline(None, 0, 1)
        run()
        torch.cuda.synchronize()
        nb = 512
        host = np.zeros((nb, 16), dtype=np.uint64)
        assert lib.sn_debug_timeline(host.ctypes.data_as(vp), nb, 0) == 0
        valid = host[:, 0] > 0
        t = host.astype(np.float64) / 100.0
        t0 = t[valid, 0].min()
        last = max(sidx for sidx in range(7) if (host[valid, sidx] > 0).all())
        print("== last GEMM layer of the %d-layer stack (%d -> %d channels): %d workgroups, span %.2f us" %
              (nl, chans[-2], chans[-1], valid.sum(), t[valid, last].max() - t0))
        for sidx in range(7):
            if (host[valid, sidx] > 0).all():
                col = t[valid, sidx] - t0
                print("     slot %d   abs med %.2f  (p10 %.2f p90 %.2f)" % (sidx, np.median(col), np.percentile(col, 10), np.percentile(col, 90)))


FUSED = True

if __name__ == "__main__":
    lib.sn_debug_timeline.argtypes = [vp, i, i]
    lib.sn_linear_forward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, vp]
    lib.sn_linear_backward.argtypes = [i, i, i, i, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, vp]
    R = int(os.environ.get("TL_BATCH", "32")) * 1024
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        small_fwd(32, 256, 256)
        small_fwd(32, 128, 256)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stack":
        stack_fwd(B=int(sys.argv[2]) if len(sys.argv) > 2 else 32)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fwd":
        fwd(R, 64, 64)
        fwd(R, 64, 128)
        fwd(R, 128, 128)
        sys.exit(0)
    if len(sys.argv) < 2:
        fwd(R, 64, 64)
        fwd(R, 64, 128)
        fwd(R, 128, 128)
    bwd(R, 64, 64, 1, B=R // 1024)
    bwd(R, 64, 128, 1, B=R // 1024)
    bwd(R, 128, 128, 2, B=R // 1024)
