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
  * the PointNet feature extractor runs through samplenet_amd.pointnet (hand-written fp32 MFMA kernels); the
    parameters stay ordinary nn.Conv1d / nn.BatchNorm1d / nn.Linear members, so checkpoints are interchangeable.
"""
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import ops, pointnet, sputils, surface
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
        conv_widths=(64, 64, 64, 128),
        fc_widths=(256, 256, 256),
        fc_batchnorm=True,
        last_fc_batchnorm=False,
        temperature_floor=None,
    ):
        """Arguments up to skip_projection: registration/src/samplenet.py:23-37.  The keyword-only-in-spirit rest selects the
        sampler variants of the TF packages (same layer names, so the registration defaults stay state_dict compatible):
          reconstruction (reconstruction/src/samplers.py:23-38, soft_projection.py:51-54):
              conv_widths=(64, 128, 128, 256), fc_widths=(256, 256), fc_batchnorm=False, temperature_floor=1e-2, min_sigma=0
          classification (classification/models/samplenet_model.py:30-108, soft_projection.py:41):
              last_fc_batchnorm=True, min_sigma=0
        """
        super().__init__()
        self.use_hip_mlp = True  # the feature extractor runs on the hand-written MFMA kernels (there is no other route here)
        self.num_out_points = num_out_points
        self.name = "samplenet"

        # Layers registered under the reference's names and in its order (samplenet.py:40-59): checkpoints are
        # interchangeable.  Default widths: 3 -> 64 -> 64 -> 64 -> 128 -> bottleneck over the points, then 256 -> 256 -> 256 -> 3*M.
        if len(conv_widths) != 4:
            raise ValueError("conv_widths: four hidden widths (the fifth conv layer ends in bottleneck_size)")
        conv_widths = (3,) + tuple(int(w) for w in conv_widths) + (bottleneck_size,)
        fc_widths = (bottleneck_size,) + tuple(int(w) for w in fc_widths) + (3 * num_out_points,)
        self.num_fc_layers = len(fc_widths) - 1
        for i in range(1, len(conv_widths)):
            self.add_module("conv%d" % i, nn.Conv1d(conv_widths[i - 1], conv_widths[i], kernel_size=1))
        for i in range(1, len(conv_widths)):
            self.add_module("bn%d" % i, nn.BatchNorm1d(conv_widths[i]))
        for i in range(1, len(fc_widths)):
            self.add_module("fc%d" % i, nn.Linear(fc_widths[i - 1], fc_widths[i]))
        if fc_batchnorm:
            for i in range(1, len(fc_widths) - 1):
                self.add_module("bn_fc%d" % i, nn.BatchNorm1d(fc_widths[i]))
        if last_fc_batchnorm:
            self.add_module("bn_fc%d" % self.num_fc_layers, nn.BatchNorm1d(fc_widths[-1]))
        # the fused single-node step (engine fast path) is specialised to the registration architecture
        self.standard_arch = (conv_widths[1:5] == (64, 64, 64, 128) and fc_widths[1:-1] == (256, 256, 256) and fc_batchnorm
                              and not last_fc_batchnorm and temperature_floor is None)

        # ... and to the classification sampler: the same layers plus a BatchNorm on the head's output (the scan then reads
        # the queries instead of computing fc4 in its waves: fused_step.py)
        self.standard_arch_out_bn = (conv_widths[1:5] == (64, 64, 64, 128) and fc_widths[1:-1] == (256, 256, 256) and fc_batchnorm
                                     and last_fc_batchnorm and temperature_floor is None)
        self.project = SoftProjection(group_size, initial_temperature, is_temperature_trainable, min_sigma,
                                      temperature_floor=temperature_floor)
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
        # training steps of one configuration run on captured work behind forward() / the loss getters (surface.py: gradients are
        # written straight into .grad; the surface steps aside where autograd hooks / DistributedDataParallel may listen);
        # False: every call op by op; "force": captured even under a multi-rank process group without a FlatGradAllReducer
        self.graph_surface = True
        # False: forward() hands out COPIES of the graphs' static outputs (what a script keeps across steps keeps its values, as
        # with the reference module); True: the static tensors themselves (overwritten by the next forward of the configuration)
        self.surface_static_outputs = False
        self.device_matching = True  # eval branch: nn_matching / FPS completion on the GPU (False: numpy, as the reference)

    # per-step / per-attachment state that must not travel with a copy of the module (graph tensors, views of another
    # module's gradient bucket, persistent kernel scratch)
    _TRANSIENT = ("_scan", "_grad_sink", "_after_fc_grads", "_colmin_keys", "_colmin_keys_owner", "_fx_acc", "_fx_acc_b",
                  "_fc_sync", "_fc_sync_b", "_sn_layer_records", "_sn_plans", "_sn_sync_bn", "_sn_surface", "_sn_surface_live",
                  "_sn_hook_params", "_sn_surface_warned", "_sn_pinned", "_sn_variant_ok")

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() ...: parameter storage moves -- recorded pointer arrays, captured graphs and persistent
        # scratch are void
        for k in ("_sn_plans", "_sn_layer_records", "_sn_surface", "_sn_surface_live", "_sn_hook_params"):
            self.__dict__.pop(k, None)
        return super()._apply(fn, *args, **kwargs)

    def __getstate__(self):
        # pickling (torch.save(module)): graphs, pointer arrays and views of foreign buckets stay behind
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._scan = None

    def __deepcopy__(self, memo):
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._TRANSIENT:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._scan = None
        return new

    # ------------------------------------------------------------------------------------------ MLP
    def _features(self, x, x_bnc=None):
        """PointNet feature extractor + FC head: x (B,3,N) -> y (B,3,M)   (samplenet.py:90-104)."""
        if x_bnc is None:
            x_bnc = x.permute(0, 2, 1)
        return pointnet.pointnet_head(self, x_bnc)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor):
        """x in `input_shape` -> (simplified cloud, projected cloud [train] | matched cloud [eval]) in `output_shape`
        (samplenet.py:85-142).  Internally the head's output y is (B,3,M); the cloud is used in whichever layout it came."""
        if self.training:
            out = surface.try_forward(self, x)  # captured forward of this configuration (None: op by op below)
            if out is None and self.__dict__.get("_sn_surface_live"):
                surface._demote_lives(self)
            if out is not None:
                self._scan = None
                return out
        cloud_is_bnc = self.input_shape == "bnc"
        out_is_bnc = self.output_shape == "bnc"
        x_bcn = x.permute(0, 2, 1) if cloud_is_bnc else x
        if x_bcn.shape[1] != 3:
            raise RuntimeError("shape of x must be of [Batch x 3 x NumInPoints]")
        y = self._features(x_bcn, x if cloud_is_bnc else None)  # (B,3,M)
        simp = (y.permute(0, 2, 1) if out_is_bnc else y).contiguous()
        self._scan = None
        if self.training:
            second = self._project_training(x, x_bcn, y, simp, cloud_is_bnc, out_is_bnc)
        else:
            second = self._match_inference(x_bcn, y, out_is_bnc)
        return simp, second

    def _project_training(self, x, x_bcn, y, simp, cloud_is_bnc, out_is_bnc):
        if self.skip_projection:
            return simp
        # one pair scan: projection + both Chamfer directions; it reads the cloud and writes the projection in either
        # layout, so nothing is transposed.  The Chamfer products are remembered for get_simplification_loss().
        cloud = (x if cloud_is_bnc else x_bcn).contiguous()
        proj, _idx, dq, iq, dp, ip = self.project.project_with_chamfer(
            cloud, y.contiguous(), ops.BNC if cloud_is_bnc else ops.BCN, ops.BNC if out_is_bnc else ops.BCN)
        self._scan = (x, x._version, simp, (dq, iq, dp, ip), y)
        return proj.contiguous()

    def _match_inference(self, x_bcn, y, out_is_bnc):
        """Nearest input point of every generated point, duplicates replaced by farthest-point picks (samplenet.py:119-141)."""
        M = self.num_out_points
        idx, _ = ops.knn(1, x_bcn.contiguous(), y.contiguous(), ops.BCN, ops.BCN, return_dist=False)  # (B,M,1)
        if self.device_matching and M <= 1024 and x_bcn.shape[2] <= 8192:
            # unique + farthest-point completion on the device (same points as sputils.nn_matching)
            match = ops.nn_matching(x_bcn.contiguous(), idx, M, self.complete_fps, ops.BCN)
        else:  # host round trip through numpy, as the reference does it
            pts = x_bcn.permute(0, 2, 1).detach().cpu().numpy()
            picked = sputils.nn_matching(pts, idx.squeeze(2).cpu().numpy(), M, complete_fps=self.complete_fps)
            match = torch.as_tensor(picked, dtype=torch.float32).to(x_bcn.device)
        return (match if out_is_bnc else match.permute(0, 2, 1)).contiguous()  # match is (B,M,3)

    def check(self):
        """Host-side health check of the FC chain launches' hand-off state (pointnet.check_chain_errors): raises
        SampleNetHipError when a step since the last check ran on incomplete data (its outputs / gradients were NaN-poisoned on
        the device already) and re-arms the launch state.  Synchronises the device: call it where the loss is read back."""
        return pointnet.check_chain_errors(self)

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
        if self.__dict__.get("_sn_surface_live"):
            loss = surface.simplification_loss(self, ref_pc, samp_pc, gamma + delta * pc_size)
            if loss is not None:  # (an output of the captured forward's own autograd node)
                return loss
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
        if self.__dict__.get("_sn_surface_live") and self.training and not self.skip_projection:
            sigma = surface.projection_loss(self)
            if sigma is not None:
                return sigma
        sigma = self.project.sigma()
        if self.skip_projection or not self.training:
            return torch.tensor(0).to(sigma)
        return sigma
