"""Test helper: the sampler's feature extractor through plain torch.nn modules (the reference's own op chain,
registration/src/samplenet.py:90-104) on the SAME parameters -- the fp32 / fp64 yardstick the HIP MLP kernels are compared
with.  Lives in tests/ on purpose: the product module has no torch route."""
import copy

import torch.nn.functional as F

from samplenet_amd import SampleNet


class TorchMLPSampleNet(SampleNet):
    def _features(self, x, x_bnc=None):
        y = x
        for i in range(1, 6):
            y = F.relu(getattr(self, "bn%d" % i)(getattr(self, "conv%d" % i)(y)))
        y = y.max(dim=2).values  # (B, bottleneck)
        nfc = self.num_fc_layers
        for i in range(1, nfc):
            y = getattr(self, "fc%d" % i)(y)
            bn = getattr(self, "bn_fc%d" % i, None)
            y = F.relu(bn(y) if bn is not None else y)
        y = getattr(self, "fc%d" % nfc)(y)
        bn = getattr(self, "bn_fc%d" % nfc, None)  # classification variant: BatchNorm on the head's output
        if bn is not None:
            y = bn(y)
        return y.view(-1, 3, self.num_out_points)


def torch_mlp_copy(net):
    """Deep copy of a SampleNet whose feature extractor runs through torch.nn (gradients through autograd)."""
    ref = copy.deepcopy(net)
    ref.__class__ = TorchMLPSampleNet
    ref.use_hip_mlp = False
    ref.__dict__.pop("_grad_sink", None)
    return ref
