"""ChamferDistance -- drop-in for registration/src/chamfer_distance/chamfer_distance.py.

Same surface (ChamferDistanceFunction.apply(xyz1, xyz2) -> dist1, dist2; ChamferDistance()(xyz1, xyz2)),
but both directions come out of ONE pair-scan launch (sn_chamfer_forward) instead of two kernels
plus four pageable host->device copies (chamfer_distance.py:21-34), and the backward is a
deterministic kernel pair (sn_chamfer_backward) instead of memset + float atomics.
"""
import torch

from . import ops


class ChamferDistanceFunction(torch.autograd.Function):
    """Two-output form of the reference function (chamfer_distance.py:14-61)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2, dist1, idx1, dist2, idx2 = ops.chamfer_forward_impl(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, graddist1, graddist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        return ops.chamfer_backward_impl(xyz1, xyz2, idx1, idx2, graddist1, graddist2, ctx.needs_input_grad[0],
                                         ctx.needs_input_grad[1])


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)
