"""pointnet2_utils.grouping_operation(features (B,C,N), idx (B,npoint,nsample) int32) -> (B,C,npoint,nsample),
differentiable w.r.t. features -- the one function of Pointnet2_PyTorch the SampleNet hot path calls
(soft_projection.py:8,86,88)."""
from .... import ops


def grouping_operation(features, idx):
    return ops.grouping_operation(features, idx)
