"""Host-side helpers that the train scripts import from `src.sputils` -- same public names and behaviour as
registration/src/sputils.py (`nn_matching` :31-41 with its farthest-point completion :7-28, `get_parser` :45-61), written
independently.  The GPU version of the matching is `samplenet_amd.ops.nn_matching` (sn_nn_matching); this numpy one is the
host fallback of the eval branch and the comparison point of the tests.
"""
import argparse

import numpy as np

__all__ = ["nn_matching", "get_parser"]


def _sqdist_to(point, cloud):
    """float64 squared distance of every cloud point to `point`, accumulated x, y, z in that order."""
    delta = np.asarray(point, dtype=np.float64)[None, :] - cloud
    return (delta[:, 0] * delta[:, 0] + delta[:, 1] * delta[:, 1]) + delta[:, 2] * delta[:, 2]


def _first_occurrences(indices):
    """The distinct values of `indices` in the order in which they first appear."""
    seen, kept = set(), []
    for v in np.asarray(indices).tolist():
        if v not in seen:
            seen.add(v)
            kept.append(v)
    return np.asarray(kept, dtype=np.asarray(indices).dtype)


def _complete_by_farthest_points(cloud, seeds, k):
    """`seeds` (t,3), then k - t more points of `cloud`, each the one farthest from everything chosen so far
    (first maximum on ties) -- the reference's _fps_from_given_pc."""
    chosen = np.zeros((k, 3), dtype=np.float64)
    t = int(np.size(seeds) // 3)
    chosen[:t] = seeds
    nearest = _sqdist_to(chosen[0], cloud)
    for i in range(1, k):
        if i >= t:
            chosen[i] = cloud[int(np.argmax(nearest))]
        nearest = np.minimum(nearest, _sqdist_to(chosen[i], cloud))
    return chosen


def nn_matching(full_pc, idx, k, complete_fps=True):
    """full_pc (B,N,3), idx (B,k) nearest-input-point indices -> (B,k,3) float64 matched points.
    complete_fps: duplicates are dropped (first occurrence kept) and the set is refilled to k points by farthest-point
    sampling of the same cloud; otherwise a plain gather."""
    full_pc = np.asarray(full_pc)
    idx = np.asarray(idx)
    matched = np.zeros((full_pc.shape[0], k, 3), dtype=np.float64)
    for b, (cloud, picks) in enumerate(zip(full_pc, idx)):
        if complete_fps:
            matched[b] = _complete_by_farthest_points(cloud, cloud[_first_occurrences(picks)], k)
        else:
            matched[b] = cloud[picks][:k]
    return matched


# (flags, keyword arguments) of every option of the reference parser, registration/src/sputils.py:45-61
_SAMPLER_OPTIONS = (
    (("--skip-projection",), dict(action="store_true", help="train without the soft projection")),
    (("-in", "--num-in-points"), dict(type=int, default=1024, help="points per input cloud [1024]")),
    (("-out", "--num-out-points"), dict(type=int, default=64, help="points per sampled cloud, 2..1024 [64]")),
    (("--bottleneck-size",), dict(type=int, default=128, help="width of the sampler's global feature [128]")),
    (("--alpha",), dict(type=float, default=0.01, help="weight of the simplification loss [0.01]")),
    (("--gamma",), dict(type=float, default=1, help="constant weight of the reverse Chamfer term [1]")),
    (("--delta",), dict(type=float, default=0, help="per-output-point weight of the reverse Chamfer term [0]")),
    (("-gs", "--projection-group-size"), dict(type=int, default=8, help="neighbours used by the soft projection [8]")),
    (("--lmbda",), dict(type=float, default=0.01, help="weight of the projection (temperature) loss [0.01]")),
)


def get_parser():
    """ArgumentParser carrying the sampler options of the reference (same flags, types and defaults)."""
    parser = argparse.ArgumentParser("SampleNet: Differentiable Point Cloud Sampling")
    for flags, kwargs in _SAMPLER_OPTIONS:
        parser.add_argument(*flags, **kwargs)
    return parser
