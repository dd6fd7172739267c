"""Progressive SampleNet for the PyTorch surface.

The reference ships the progressive sampler only as TF graphs (classification/train_samplenet_progressive.py:157-234,
reconstruction/src/samplenet_progressive_pointnet_ae.py:77-100,165-173); its semantics:
  * the sampler emits the LARGEST set once (MAX_NUM_OUT_POINTS points), the whole set is soft-projected once;
  * every nested size s in {min, 2 min, ..., max} uses the PREFIX [:s] of the simplified / projected points:
    the task network is fed projected[:, :s], the simplification loss of size s is taken on simplified[:, :s]
    with pc_size = s (its delta term);
  * classification: total simplification loss = SUM over sizes; auto-encoder variant: MEAN over sizes.
This module is that layer over the HIP hot path: SampleNetProgressive is a SampleNet whose losses take a list of sizes.
The query->point products (dist / idx per simplified point) do not depend on the prefix, so they come from forward()'s
pair scan; the point->query side of EVERY prefix comes from one more pass over the distances (sn_prefix_point_minima:
running minimum over the queries in index order, emitted at each prefix end).
"""
import torch

from . import ops, surface
from .samplenet import SampleNet


def progressive_sizes(min_points, max_points):
    """{min, 2 min, 4 min, ... <= max} as classification/train_samplenet_progressive.py:181-199 builds it."""
    if min_points < 1 or max_points < min_points:
        raise ValueError("need 1 <= min_points <= max_points")
    sizes, b = [], min_points
    while b <= max_points:
        sizes.append(b)
        b *= 2
    return sizes


class SampleNetProgressive(SampleNet):
    """SampleNet(num_out_points = the largest size) + prefix-wise losses.

    forward(x) -> (simp, proj) exactly as SampleNet; slice both with [:, :s] (output_shape 'bnc') for the task network.
    """

    def __init__(self, sizes, bottleneck_size, group_size, **kw):
        sizes = sorted(int(s) for s in sizes)
        if not sizes or sizes[0] < 1 or len(set(sizes)) != len(sizes):
            raise ValueError("sizes must be distinct positive integers")
        super().__init__(sizes[-1], bottleneck_size, group_size, **kw)
        self.sizes = sizes
        self.name = "samplenet_progressive"
        # the prefix losses hang off slices of the simplified cloud: the captured surface takes their gradient as an operand of its
        # backward graph (surface.py: with_simp) from the first capture on
        self.__dict__["_sn_surface_simp_grad"] = True

    def prefix(self, pc, size):
        """The first `size` points of a (B,M,3) ['bnc'] or (B,3,M) ['bcn'] cloud in the module's output layout."""
        return pc[:, :size, :] if self.output_shape == "bnc" else pc[:, :, :size]

    def prefixes(self, pc, sizes=None):
        """[prefix(pc, s) for s in sizes] (default: the module's sizes) as CONTIGUOUS tensors from one launch -- and their gradients
        come back summed in one launch (ops.prefix_pack) -- for the 'bnc' output layout on the GPU; the plain slices otherwise."""
        sizes = self.sizes if sizes is None else list(sizes)
        if self.output_shape == "bnc" and pc.is_cuda and pc.dim() == 3 and pc.dtype == torch.float32 and len(sizes) <= 16:
            return ops.prefix_pack(pc, sizes)
        return [self.prefix(pc, s) for s in sizes]

    def get_progressive_simplification_loss(self, ref_pc, samp_pc, gamma=1, delta=0, reduction="sum"):
        """sum (classification) or mean (auto-encoder) over the sizes of
        get_simplification_loss(ref_pc, samp_pc[:, :s], s, gamma, delta)      [ref_pc, samp_pc: (B,N,3), (B,M,3)]"""
        if reduction not in ("sum", "mean"):
            raise ValueError("reduction must be 'sum' or 'mean'")
        if self.skip_projection or not self.training:
            return torch.tensor(0).to(ref_pc)
        # One pass over the M x N distances serves every prefix: the per-query products (dist1 / idx1) do not
        # depend on the prefix -- they are forward()'s scan (or one Chamfer scan) sliced -- and the per-point products of all
        # prefixes come from ops.prefix_point_minima's running minimum in one launch.
        M = samp_pc.shape[1]
        if self.sizes[-1] != M:
            raise ValueError("samp_pc must hold the largest size (%d points)" % self.sizes[-1])
        scan = self._scan_hit(ref_pc, samp_pc)
        if scan is not None:
            dq, iq = scan[0], scan[1]
        else:
            _, _, dq, iq, _, _ = ops.chamfer_forward_impl(samp_pc.detach(), ref_pc.detach())
        d2, i2 = ops.prefix_point_minima(ref_pc, samp_pc, self.sizes)
        # the terms of the proper prefixes as one autograd node (two launches forward, one backward, no copies of the prefixes);
        # a reference cloud that takes a gradient itself, or more than 2048 points: term by term on contiguous copies
        part = [s for s in self.sizes if s != M]
        fused = bool(part) and samp_pc.is_cuda and not ref_pc.requires_grad and ref_pc.shape[1] <= 2048 and len(part) <= 16
        total = None
        if fused:
            total = ops.PrefixSimplificationLossFunction.apply(samp_pc, ref_pc, dq, iq, d2, i2, part, [gamma + delta * s for s in part])
        for j, s in enumerate(self.sizes):
            if fused and s != M:
                continue
            live = None
            if s == M and self.__dict__.get("_sn_surface_live"):  # the full size: the captured forward's own L_simp output
                live = surface.simplification_loss(self, ref_pc, samp_pc, gamma + delta * s)
            if live is not None:
                term = live
            elif s == M and scan is not None and not ref_pc.requires_grad:
                term = self.get_simplification_loss(ref_pc, samp_pc, s, gamma, delta)  # hangs off the head's (B,3,M) output
            else:
                sl = samp_pc[:, :s, :].contiguous()
                term = ops.SimplificationLossFunction.apply(sl, ref_pc, dq[:, :s].contiguous(), iq[:, :s].contiguous(), d2[j], i2[j],
                                                            gamma + delta * s)
            total = term if total is None else total + term
        return total / len(self.sizes) if reduction == "mean" else total
