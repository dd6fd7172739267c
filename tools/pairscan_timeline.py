#!/usr/bin/env python3
"""Phase timelines of the pair-scan kernel and of the loss backward behind it (chamfer_soft_bwd_kernel) as the engine
launches them (queries produced by fc4 inside the scan, per-point minima combined by atomicMax on keys), from a debug build
with -DSN_PS_TIMELINE=L -DSN_CS_TIMELINE=L, L = 1 (stamps only) or 2 (every stamp first waits for the memory operations
before it, i.e. serialised phases):

    library built with those flags (pairscan.hip, geometry_ops.hip);   python tools/pairscan_timeline.py <that library> [B]

Thread 0 of every workgroup stamps the 100 MHz wall clock; printed: median / p10 / p90 over the workgroups of the time
between consecutive stamps, in microseconds."""
import ctypes
import sys

import numpy as np
import torch

lib = ctypes.CDLL(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, M, K, KF = 1024, 64, 8, 256
vp = ctypes.c_void_p


def P(t):
    return None if t is None else vp(t.data_ptr())


dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
z3 = torch.randn(B, KF, device=dev, generator=g)
scale, shift = torch.ones(KF, device=dev), torch.zeros(KF, device=dev)
W = torch.randn(3 * M, KF, device=dev, generator=g) * 0.02
bias = torch.zeros(3 * M, device=dev)
Q = torch.empty(B, 3, M, device=dev)
idx = torch.empty(B, M, K, device=dev, dtype=torch.int32)
dq, iq = torch.empty(B, M, device=dev), torch.empty(B, M, device=dev, dtype=torch.int32)
proj = torch.empty(B, M, 3, device=dev)
T = torch.ones(1, device=dev)
lib.sn_pairscan_colmin_splits.restype = ctypes.c_int
G = lib.sn_pairscan_colmin_splits(B, N, M)
keys = torch.zeros(B, N, device=dev, dtype=torch.int64)
qpart = torch.empty(B, G, 2, device=dev)
qmax = torch.empty(B, G, device=dev, dtype=torch.int64)
st = vp(torch.cuda.current_stream().cuda_stream)
nblocks = B * G


def launch():
    keys.zero_()
    rc = lib.sn_pairscan_forward_keys(B, N, M, K, P(x), 0, P(Q), P(z3), P(scale), P(shift), P(W), P(bias), KF, P(idx), P(dq), P(iq),
                                      P(proj), 0, P(T), ctypes.c_float(1e-2), P(keys), P(qpart), P(qmax), st)
    assert rc == 0, rc


for _ in range(3):
    launch()
torch.cuda.synchronize()
host = np.zeros((nblocks, 16), dtype=np.uint64)
assert lib.sn_debug_pairscan_timeline(host.ctypes.data_as(vp), nblocks) == 0  # clears the warm-up stamps
launch()
assert lib.sn_debug_pairscan_timeline(host.ctypes.data_as(vp), nblocks) == 0
t = host.astype(np.float64) / 100.0
names = ["start", "cloud in registers", "query ready (fc4)", "distances + threshold", "candidates in LDS", "ranked",
         "query outputs done", "column minima in LDS", "keys combined (global atomics)", "end"]
t0 = t[:, 0].min()
print("B = %d: %d workgroups (%d per cloud); first start -> last end %.2f us; start spread p90 %.2f us" %
      (B, nblocks, G, t[:, 9].max() - t0, np.percentile(t[:, 0] - t0, 90)))
for s in range(1, 10):
    d = t[:, s] - t[:, s - 1]
    print("  %-32s +%.2f  (p10 %.2f  p90 %.2f)   at %.2f" % (names[s], np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                            np.median(t[:, s] - t0)))

# ---- the backward of the loss side behind it
if hasattr(lib, "sn_debug_chamfer_soft_bwd_timeline"):
    lib.sn_soft_bwd_splits.restype = ctypes.c_int
    splits = min(max(1, min((M + 3) // 4, (512 + B - 1) // B)), lib.sn_soft_bwd_splits(B, M))
    gl = torch.ones(1, device=dev)
    gQ = torch.zeros(B, 3, M, device=dev)
    gsig = torch.zeros(B * 64, device=dev)
    gT, dps, loss = torch.zeros(1, device=dev), torch.zeros(B, device=dev), torch.zeros(2, device=dev)
    cf = ctypes.c_float

    def bwd():
        launch()
        rc = lib.sn_sampler_step_loss_keys(B, N, M, K, P(x), 0, P(Q), P(idx), P(iq), P(keys), P(qpart), P(qmax), G, P(T), cf(1e-2),
                                           cf(0.01), cf(0.01), cf(1.0), P(gl), P(gQ), P(gsig), P(gT), P(dps), P(loss), st, None, None, None)
        assert rc == 0, rc

    for _ in range(3):
        bwd()
    nb = B * splits
    h2 = np.zeros((nb, 16), dtype=np.uint64)
    assert lib.sn_debug_chamfer_soft_bwd_timeline(h2.ctypes.data_as(vp), nb) == 0
    bwd()
    assert lib.sn_debug_chamfer_soft_bwd_timeline(h2.ctypes.data_as(vp), nb) == 0
    t = h2.astype(np.float64) / 100.0
    names = ["start", "keys -> LDS, barrier", "cloud in registers", "query + own term loaded", "matches accumulated",
             "soft projection backward", "end"]
    t0 = t[:, 0].min()
    print("chamfer_soft_bwd_kernel: %d workgroups (%d per cloud); first start -> last end %.2f us; start spread p90 %.2f us" %
          (nb, splits, t[:, 6].max() - t0, np.percentile(t[:, 0] - t0, 90)))
    if (h2[:, 8] > 0).all():
        print("  inside the first phase: cloud / scalars / first query requested and landed +%.2f, keys and dependent gathers landed "
              "+%.2f, keys -> LDS and reductions +%.2f, barrier +%.2f" %
              tuple(np.median(t[:, b_] - t[:, a_]) for a_, b_ in ((0, 8), (8, 9), (9, 10), (10, 1))))
    if (h2[:, 11] > 0).all():
        print("  (keys alone landed +%.2f after the first batch, dependent gathers +%.2f after them)" %
              (np.median(t[:, 11] - t[:, 8]), np.median(t[:, 9] - t[:, 11])))
    for s_ in range(1, 7):
        d = t[:, s_] - t[:, s_ - 1]
        print("  %-32s +%.2f  (p10 %.2f  p90 %.2f)   at %.2f" % (names[s_], np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                                np.median(t[:, s_] - t0)))
