"""`cd` -- the native plugin surface of the reference's Chamfer distance, on libsamplenet_hip.so.

The reference builds a pybind11 module named `cd` at import time (registration/src/chamfer_distance/chamfer_distance.py:5-11,
`torch.utils.cpp_extension.load(name="cd", sources=[chamfer_distance.cpp, chamfer_distance.cu])`) that exports four functions
(chamfer_distance.cpp:180-185):

    forward(xyz1, xyz2, dist1, dist2, idx1, idx2)                                   CPU loop
    forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)                              ChamferDistanceKernelLauncher
    backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)      CPU loop
    backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2) ChamferDistanceGradKernelLauncher

all on tensors the Python caller has ALLOCATED (outputs are written in place, the functions return nothing).  This module is
that surface over the C ABI (sn_chamfer_forward / sn_chamfer_backward: same launcher signatures + a stream): replace lines 5-11
of the reference file by `from samplenet_amd.compat import cd` and the rest of it -- ChamferDistanceFunction.forward /
backward, output allocation, save_for_backward -- runs unchanged (level 0).  Differences: the launches go to
torch's CURRENT stream (the reference launchers use the legacy default stream), errors raise instead of being printed
(chamfer_distance.cu:152-154), and there is no CPU path (`forward` / `backward` raise).
"""
import torch

from .._lib import check, lib, ptr


def _dev(name, t, dtype):
    if not t.is_cuda:
        raise RuntimeError("cd.%s: %s must be a GPU tensor (samplenet_amd has no CPU path)" % (name, "every tensor"))
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError("cd.%s: tensors must be contiguous %s (chamfer_distance.py:19-20,44-45)" % (name, dtype))
    return t


def forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """xyz1 (B,n,3), xyz2 (B,m,3) float32; writes dist1 (B,n), idx1 (B,n) int32, dist2 (B,m), idx2 (B,m) int32 in place."""
    for t in (xyz1, xyz2, dist1, dist2):
        _dev("forward_cuda", t, torch.float32)
    for t in (idx1, idx2):
        _dev("forward_cuda", t, torch.int32)
    b, n, m = xyz1.size(0), xyz1.size(1), xyz2.size(1)
    if dist1.numel() != b * n or idx1.numel() != b * n or dist2.numel() != b * m or idx2.numel() != b * m:
        raise RuntimeError("cd.forward_cuda: output tensors do not match the clouds' shapes")
    with torch.cuda.device(xyz1.device):
        check(lib.sn_chamfer_forward(b, n, ptr(xyz1), m, ptr(xyz2), ptr(dist1), ptr(idx1), ptr(dist2), ptr(idx2),
                                     torch.cuda.current_stream(xyz1.device).cuda_stream), "sn_chamfer_forward")


def backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """Writes gradxyz1 (B,n,3) / gradxyz2 (B,m,3) in place (overwritten: the caller's zero fill is not needed)."""
    for t in (xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2):
        _dev("backward_cuda", t, torch.float32)
    for t in (idx1, idx2):
        _dev("backward_cuda", t, torch.int32)
    b, n, m = xyz1.size(0), xyz1.size(1), xyz2.size(1)
    if gradxyz1.shape != xyz1.shape or gradxyz2.shape != xyz2.shape:
        raise RuntimeError("cd.backward_cuda: gradient tensors do not match the clouds' shapes")
    with torch.cuda.device(xyz1.device):
        check(lib.sn_chamfer_backward(b, n, ptr(xyz1), m, ptr(xyz2), ptr(graddist1), ptr(idx1), ptr(graddist2), ptr(idx2),
                                      ptr(gradxyz1), ptr(gradxyz2), torch.cuda.current_stream(xyz1.device).cuda_stream),
              "sn_chamfer_backward")


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    raise RuntimeError("cd.forward is the reference's CPU loop (chamfer_distance.cpp:112-131); samplenet_amd has no CPU path -- "
                       "move the tensors to the GPU (forward_cuda)")


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    raise RuntimeError("cd.backward is the reference's CPU loop (chamfer_distance.cpp:133-177); samplenet_amd has no CPU path -- "
                       "move the tensors to the GPU (backward_cuda)")
