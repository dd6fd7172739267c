"""knn_cuda.KNN with the interface the reference uses (soft_projection.py:11-14, samplenet.py:121):
KNN(k, transpose_mode=False)(ref (B,C,N), query (B,C,M)) -> dist (B,k,M) Euclidean, idx (B,k,M) int64
(transpose_mode=True: ref (B,N,C), query (B,M,C) -> (B,M,k)).  Neighbours ascend by (distance, index)."""
import torch

from .. import ops


class KNN(torch.nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        lay = ops.BNC if self._t else ops.BCN
        if ref.shape[2 if self._t else 1] != 3:
            raise NotImplementedError("samplenet_amd KNN handles 3-D points (the SampleNet hot path)")
        idx, d2 = ops.knn(self.k, ref, query, lay, lay, return_dist=True)
        d, idx = d2.sqrt(), idx.long()
        if not self._t:
            d, idx = d.permute(0, 2, 1).contiguous(), idx.permute(0, 2, 1).contiguous()
        return d, idx
