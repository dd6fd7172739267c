"""Host-side utilities shared with the train scripts -- same names and behaviour as
registration/src/sputils.py (nn_matching :31-41, get_parser :45-61), re-implemented.
"""
import argparse

import numpy as np


def _calc_distances(p0, points):
    return ((p0 - points) ** 2).sum(axis=1)


def _fps_from_given_pc(pts, k, given_pc):
    """Farthest-point completion of `given_pc` to k points drawn from pts (sputils.py:11-23)."""
    out = np.zeros((k, 3))
    t = np.size(given_pc) // 3
    out[0:t] = given_pc
    dist = _calc_distances(out[0], pts)
    for i in range(1, t):
        dist = np.minimum(dist, _calc_distances(out[i], pts))
    for i in range(t, k):
        out[i] = pts[np.argmax(dist)]
        dist = np.minimum(dist, _calc_distances(out[i], pts))
    return out


def _unique(arr):
    """Unique values in first-occurrence order (sputils.py:26-28)."""
    _, first = np.unique(arr, return_index=True)
    return arr[np.sort(first)]


def nn_matching(full_pc, idx, k, complete_fps=True):
    """full_pc (B,N,3), idx (B,k) -> matched points (B,k,3) (sputils.py:31-41)."""
    batch_size = np.size(full_pc, 0)
    out_pc = np.zeros((full_pc.shape[0], k, 3))
    for ii in range(batch_size):
        best_idx = idx[ii]
        if complete_fps:
            best_idx = _unique(best_idx)
            out_pc[ii] = _fps_from_given_pc(full_pc[ii], k, full_pc[ii][best_idx])
        else:
            out_pc[ii] = full_pc[ii][best_idx]
    return out_pc[:, 0:k, :]


# fmt: off
def get_parser():
    """Argument parser with exactly the flags of registration/src/sputils.py:45-61."""
    parser = argparse.ArgumentParser("SampleNet: Differentiable Point Cloud Sampling")
    parser.add_argument("--skip-projection", action="store_true", help="Do not project points in training")
    parser.add_argument("-in", "--num-in-points", type=int, default=1024, help="Number of input Points [default: 1024]")
    parser.add_argument("-out", "--num-out-points", type=int, default=64, help="Number of output points [2, 1024] [default: 64]")
    parser.add_argument("--bottleneck-size", type=int, default=128, help="bottleneck size [default: 128]")
    parser.add_argument("--alpha", type=float, default=0.01, help="Simplification regularization loss weight [default: 0.01]")
    parser.add_argument("--gamma", type=float, default=1, help="Lb constant regularization loss weight [default: 1]")
    parser.add_argument("--delta", type=float, default=0, help="Lb linear regularization loss weight [default: 0]")
    parser.add_argument("-gs", "--projection-group-size", type=int, default=8, help="Neighborhood size in Soft Projection [default: 8]")
    parser.add_argument("--lmbda", type=float, default=0.01, help="Projection regularization loss weight [default: 0.01]")
    return parser
# fmt: on
