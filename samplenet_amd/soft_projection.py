"""SoftProjection -- drop-in for registration/src/soft_projection.py (same constructor, same
`forward(point_cloud, query_cloud, point_features=None, action=...)`, same `_temperature` parameter
name, same `sigma()`), running on the HIP kernels of libsamplenet_hip.so.

  project               : one fused kernel (kNN + softmax + weighted sum), fused backward
  propagate / project_and_propagate : kNN kernel + softmax-weights kernel + weighted-gather kernel(s)
"""
import torch
import torch.nn as nn

from . import ops


def knn_point(group_size, point_cloud, query_cloud):
    """(dist, idx) like knn_cuda.KNN(k, transpose_mode=False): dist (B,k,M) Euclidean, idx (B,k,M) int64.
    Mirrors registration/src/soft_projection.py:11-14."""
    idx, d2 = ops.knn(group_size, point_cloud, query_cloud, ops.BCN, ops.BCN, return_dist=True)
    return d2.sqrt().permute(0, 2, 1).contiguous(), idx.permute(0, 2, 1).contiguous().long()


class SoftProjection(nn.Module):
    def __init__(self, group_size, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-4, temperature_floor=None):
        """Computes a soft nearest neighbor point cloud (arguments as soft_projection.py:23-44).
        temperature_floor (not in the registration module): the reconstruction package clamps the temperature itself,
        sigma = max(T, floor)^2 (reconstruction/src/soft_projection.py:51-54, floor 1e-2); None: sigma = max(T^2, min_sigma)."""
        super().__init__()
        self._temperature_floor = None if temperature_floor is None else float(temperature_floor)
        self._group_size = group_size
        self._temperature = torch.nn.Parameter(
            torch.tensor(initial_temperature, requires_grad=is_temperature_trainable, dtype=torch.float32))
        self._min_sigma = torch.tensor(min_sigma, dtype=torch.float32)  # plain attribute, as in the reference
        self._min_sigma_f = float(min_sigma)
        self._min_sigma_dev = {}  # per-device copies (the reference re-uploads it on every call)

    def forward(self, point_cloud, query_cloud, point_features=None, action="project"):
        point_cloud = point_cloud.contiguous()
        query_cloud = query_cloud.contiguous()
        if action == "project":
            return self.project(point_cloud, query_cloud)
        elif action == "propagate":
            return self.propagate(point_cloud, point_features, query_cloud)
        elif action == "project_and_propagate":
            return self.project_and_propagate(point_cloud, point_features, query_cloud)
        else:
            raise ValueError(
                "action should be one of the following: 'project', 'propagate', 'project_and_propagate'")

    def _t(self):
        """The temperature the kernels square: the parameter itself, or max(T, floor) for the reconstruction variant (a torch
        op: its gradient gate is autograd's)."""
        if self._temperature_floor is None:
            return self._temperature
        return torch.clamp(self._temperature, min=self._temperature_floor)

    def _gate_floor_(self, g):
        """In place, ONE launch: g (the gradient w.r.t. max(T, floor), one element) -> the gradient w.r.t. T, i.e. g where
        T >= floor and 0 elsewhere -- torch.clamp's own gate.  (aten's threshold_backward keeps g where self > threshold: the
        threshold is the float below the floor.)"""
        import numpy as np

        thr = float(np.nextafter(np.float32(self._temperature_floor), np.float32(-np.inf)))
        torch.ops.aten.threshold_backward.grad_input(g, self._temperature.detach().reshape(g.shape), thr, grad_input=g)
        return g

    def sigma(self):
        if self._temperature_floor is not None:
            t = self._t()
            return torch.clamp(t * t, min=self._min_sigma_f)
        if self._temperature.is_cuda and self._temperature.dtype == torch.float32 and self._temperature.numel() == 1:
            return ops.SigmaFunction.apply(self._temperature, self._min_sigma_f)  # one launch each way
        device = self._temperature.device
        ms = self._min_sigma_dev.get(device)
        if ms is None:
            ms = self._min_sigma.to(device)
            self._min_sigma_dev[device] = ms
        return torch.max(self._temperature ** 2, ms)

    # -- fused hot path -------------------------------------------------------------------------
    def project(self, point_cloud, query_cloud, hard=False):
        if hard:
            # classification/soft_projection.py:73-76: the softmax weights become one_hot(argmax) -> every query moves onto
            # its nearest input point (first of the K neighbours on ties); the registration reference raises
            # NotImplementedError here (soft_projection.py:144-145).  No gradient reaches the query (one_hot has none).
            idx, _ = ops.knn(1, point_cloud, query_cloud, ops.BCN, ops.BCN, return_dist=False)  # (B,M,1)
            return ops.grouping_operation(point_cloud.contiguous(), idx).squeeze(3)  # (B,3,M)
        proj, _idx = ops.SoftProjectFunction.apply(point_cloud, query_cloud, self._t(), self._min_sigma_f,
                                                   self._group_size, False)
        return proj

    def project_with_chamfer(self, point_cloud, query_cloud, p_layout=ops.BCN, out_layout=ops.BCN):
        """project() plus both nearest-neighbour directions between query_cloud and point_cloud from the same
        distance scan: returns proj, idx (B,M,K), dist_q (B,M), idx_q, dist_p (B,N), idx_p.
        point_cloud may be given point-major ((B,N,3), p_layout=BNC) and proj requested point-major ((B,M,3))."""
        return ops.SoftProjectFunction.apply(point_cloud, query_cloud, self._t(), self._min_sigma_f,
                                             self._group_size, True, p_layout, out_layout)

    # -- split path (features) ------------------------------------------------------------------
    def _weights(self, point_cloud, query_cloud):
        idx, _ = ops.knn(self._group_size, point_cloud, query_cloud, ops.BCN, ops.BCN, return_dist=False)
        w = ops.SoftWeightsFunction.apply(point_cloud, query_cloud, idx, self._t(), self._min_sigma_f)
        return idx, w

    def propagate(self, point_cloud, point_features, query_cloud):
        idx, w = self._weights(point_cloud, query_cloud)
        return ops.WeightedGatherFunction.apply(point_features.contiguous(), idx, w)

    def project_and_propagate(self, point_cloud, point_features, query_cloud):
        idx, w = self._weights(point_cloud, query_cloud)
        projected_points = ops.WeightedGatherFunction.apply(point_cloud, idx, w)
        propagated_features = ops.WeightedGatherFunction.apply(point_features.contiguous(), idx, w)
        return (projected_points, propagated_features)
