"""SampleNet -- drop-in for registration/src/samplenet.py (same constructor arguments, attribute and
parameter names -> state_dict compatible; forward(), get_simplification_loss(), get_projection_loss()).

What changes underneath (MI355X-native):
  * training forward: the projection (kNN -> softmax -> weighted sum) AND both Chamfer directions
    between the simplified cloud and the input cloud come from ONE fused HIP pair-scan launch;
    get_simplification_loss() then reuses those distances when it is called -- as
    registration/main.py:507-529 does -- with the very tensors forward() consumed / returned,
    instead of recomputing the distance matrix (the reference computes it three times);
  * ChamferDistance / SoftProjection / kNN run on libsamplenet_hip.so (no third-party knn_cuda /
    pointnet2 packages, no JIT-compiled extension);
  * the PointNet feature extractor runs through samplenet_amd.pointnet (hand-written MFMA kernels)
    when enabled, else through torch.nn (identical parameters either way).
"""
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, pointnet, sputils
from .chamfer_distance import ChamferDistance
from .soft_projection import SoftProjection


class SampleNet(nn.Module):
    def __init__(
        self,
        num_out_points,
        bottleneck_size,
        group_size,
        initial_temperature=1.0,
        is_temperature_trainable=True,
        min_sigma=1e-2,
        input_shape="bcn",
        output_shape="bcn",
        complete_fps=True,
        skip_projection=False,
        use_hip_mlp=True,
    ):
        super().__init__()
        # use_hip_mlp=False routes the feature extractor through torch.nn (MIOpen / rocBLAS); it exists for
        # A/B parity tests of the hand-written MFMA kernels -- the geometric ops are on HIP either way.
        self.use_hip_mlp = use_hip_mlp
        self.num_out_points = num_out_points
        self.name = "samplenet"

        # parameter names / shapes as samplenet.py:40-59 (state_dict compatibility)
        self.conv1 = torch.nn.Conv1d(3, 64, 1)
        self.conv2 = torch.nn.Conv1d(64, 64, 1)
        self.conv3 = torch.nn.Conv1d(64, 64, 1)
        self.conv4 = torch.nn.Conv1d(64, 128, 1)
        self.conv5 = torch.nn.Conv1d(128, bottleneck_size, 1)

        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(64)
        self.bn3 = nn.BatchNorm1d(64)
        self.bn4 = nn.BatchNorm1d(128)
        self.bn5 = nn.BatchNorm1d(bottleneck_size)

        self.fc1 = nn.Linear(bottleneck_size, 256)
        self.fc2 = nn.Linear(256, 256)
        self.fc3 = nn.Linear(256, 256)
        self.fc4 = nn.Linear(256, 3 * num_out_points)

        self.bn_fc1 = nn.BatchNorm1d(256)
        self.bn_fc2 = nn.BatchNorm1d(256)
        self.bn_fc3 = nn.BatchNorm1d(256)

        self.project = SoftProjection(group_size, initial_temperature, is_temperature_trainable, min_sigma)
        self.skip_projection = skip_projection
        self.complete_fps = complete_fps

        if input_shape not in ["bcn", "bnc"]:
            raise ValueError("allowed shape are 'bcn' (batch * channels * num_in_points), 'bnc' ")
        if output_shape not in ["bcn", "bnc"]:
            raise ValueError("allowed shape are 'bcn' (batch * channels * num_in_points), 'bnc' ")
        if input_shape != output_shape:
            warnings.warn("SampleNet: input_shape is different to output_shape.")
        self.input_shape = input_shape
        self.output_shape = output_shape

        self._scan = None  # Chamfer products of the last training forward (see get_simplification_loss)
        self.device_matching = True  # eval branch: nn_matching / FPS completion on the GPU (False: numpy, as the reference)

    # ------------------------------------------------------------------------------------------ MLP
    def _features(self, x, x_bnc=None):
        """PointNet feature extractor + FC head: x (B,3,N) -> y (B,3,M)   (samplenet.py:90-104)."""
        if self.use_hip_mlp:
            if x_bnc is None:
                x_bnc = x.permute(0, 2, 1)
            return pointnet.pointnet_head(self, x_bnc)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = F.relu(self.bn3(self.conv3(y)))
        y = F.relu(self.bn4(self.conv4(y)))
        y = F.relu(self.bn5(self.conv5(y)))  # Batch x bottleneck x NumInPoints
        y = torch.max(y, 2)[0]  # Batch x bottleneck
        y = F.relu(self.bn_fc1(self.fc1(y)))
        y = F.relu(self.bn_fc2(self.fc2(y)))
        y = F.relu(self.bn_fc3(self.fc3(y)))
        y = self.fc4(y)
        return y.view(-1, 3, self.num_out_points)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor):
        x_in = x
        if self.input_shape == "bnc":
            x = x.permute(0, 2, 1)
        if x.shape[1] != 3:
            raise RuntimeError("shape of x must be of [Batch x 3 x NumInPoints]")

        y = self._features(x, x_in if self.input_shape == "bnc" else None)
        simp = y
        match = None
        proj = None
        scan = None

        proj_is_out_layout = False
        if self.training:
            if not self.skip_projection:
                # the pair-scan kernel reads the cloud and writes the projection in either layout: no transposed copies
                bnc_in, bnc_out = self.input_shape == "bnc", self.output_shape == "bnc"
                proj, _idx, dq, iq, dp, ip = self.project.project_with_chamfer(
                    x_in.contiguous() if bnc_in else x.contiguous(), y.contiguous(),
                    ops.BNC if bnc_in else ops.BCN, ops.BNC if bnc_out else ops.BCN)
                proj_is_out_layout = True
                scan = (dq, iq, dp, ip)
            else:
                proj = simp
        else:  # inference: nearest-neighbour matching + FPS completion (samplenet.py:119-141)
            idx, _ = ops.knn(1, x.contiguous(), y.contiguous(), ops.BCN, ops.BCN, return_dist=False)  # (B,M,1)
            if self.device_matching and self.num_out_points <= 1024 and x.shape[2] <= 8192:
                # unique + farthest-point completion on the device (same points as sputils.nn_matching)
                match = ops.nn_matching(x.contiguous(), idx, self.num_out_points, self.complete_fps, ops.BCN)
            else:  # the reference's host round trip (samplenet.py:124-133)
                x_np = x.permute(0, 2, 1).cpu().detach().numpy()
                idx_np = idx.squeeze(2).cpu().numpy()
                z = sputils.nn_matching(x_np, idx_np, self.num_out_points, complete_fps=self.complete_fps)
                match = torch.tensor(z, dtype=torch.float32).to(x.device)  # B x M x 3

        if self.output_shape == "bnc":
            simp = simp.permute(0, 2, 1)
            if proj is not None and not proj_is_out_layout:
                proj = proj.permute(0, 2, 1)
        elif self.output_shape == "bcn" and match is not None:
            match = match.permute(0, 2, 1)
            match = match.contiguous()

        simp = simp.contiguous()
        if proj is not None:
            proj = proj.contiguous()
        if match is not None:
            match = match.contiguous()

        self._scan = None
        if scan is not None:
            # remember which tensors the scan belongs to: the input cloud and the simplified cloud we return
            self._scan = (x_in, x_in._version, simp, scan, y)

        out = proj if self.training else match
        return simp, out

    def sample(self, x):
        simp, proj = self.__call__(x)
        return proj

    # ------------------------------------------------------------------------------------------ losses
    def _scan_hit(self, ref_pc, samp_pc):
        if self._scan is None:
            return None
        x_in, ver, simp, scan, y_bcn = self._scan
        if samp_pc is not simp or samp_pc.dim() != 3 or samp_pc.shape[2] != 3:
            return None
        same_ref = ref_pc is x_in or (
            ref_pc.data_ptr() == x_in.data_ptr() and ref_pc.shape == x_in.shape and ref_pc.stride() == x_in.stride())
        if not same_ref or x_in._version != ver or ref_pc.dim() != 3 or ref_pc.shape[2] != 3:
            return None
        return scan + (y_bcn,)

    def get_simplification_loss(self, ref_pc, samp_pc, pc_size, gamma=1, delta=0):
        if self.skip_projection or not self.training:
            return torch.tensor(0).to(ref_pc)
        # ref_pc and samp_pc are B x N x 3 matrices
        scan = self._scan_hit(ref_pc, samp_pc)
        if scan is not None:  # Chamfer products of this very pair were produced by forward()'s pair scan
            dq, iq, dp, ip, y_bcn = scan
            if not ref_pc.requires_grad:
                # samp_pc is the (B,M,3) copy of the FC head's (B,3,M) output: hang the loss off that tensor directly, so
                # the gradient reaches the head without passing through the transposed copy (same values, no copy kernels)
                return ops.SimplificationLossFunction.apply(y_bcn, ref_pc, dq, iq, dp, ip, gamma + delta * pc_size, ops.BCN)
        else:
            _, _, dq, iq, dp, ip = ops.chamfer_forward_impl(samp_pc.detach(), ref_pc.detach())
        # cost_p1_p2 = mean(dq); max_cost = mean_b(max_m dq); cost_p2_p1 = mean(dp)
        # loss = cost_p1_p2 + max_cost + (gamma + delta * pc_size) * cost_p2_p1      -- one fused kernel pair
        return ops.SimplificationLossFunction.apply(samp_pc, ref_pc, dq, iq, dp, ip, gamma + delta * pc_size)

    def get_projection_loss(self):
        sigma = self.project.sigma()
        if self.skip_projection or not self.training:
            return torch.tensor(0).to(sigma)
        return sigma
