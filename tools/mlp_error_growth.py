"""Where does the HIP MLP's fp32 rounding error come from?  Per layer: pre-BatchNorm outputs of the HIP kernels, of torch
fp32 on the GPU (rocBLAS / MIOpen) and of torch fp32 on the CPU (MKL: what the golden fixtures were made with), each against
torch fp64, on the headline shape (B=32, N=1024).  Run on the GPU box: python tools/mlp_error_growth.py"""
import copy
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenet_amd import SampleNet, pointnet  # noqa: E402


def torch_layers(net, x_bcn):
    outs = []
    y = x_bcn
    for i in range(1, 6):
        z = getattr(net, "conv%d" % i)(y)
        outs.append(z)
        y = F.relu(getattr(net, "bn%d" % i)(z))
    y = y.max(dim=2).values
    for i in range(1, 4):
        z = getattr(net, "fc%d" % i)(y)
        outs.append(z)
        y = F.relu(getattr(net, "bn_fc%d" % i)(z))
    outs.append(net.fc4(y))
    return outs


def rel(a, b):
    return float((a.double().cpu() - b.cpu()).norm() / b.cpu().norm())


def main():
    torch.manual_seed(0)
    B, N = 32, 1024
    net = SampleNet(64, 128, group_size=8, input_shape="bnc", output_shape="bnc").cuda().train()
    x = torch.rand(B, N, 3, device="cuda") - 0.5
    n64 = copy.deepcopy(net).double()
    ngpu = copy.deepcopy(net)
    ncpu = copy.deepcopy(net).cpu()
    with torch.no_grad():
        o64 = torch_layers(n64, x.double().permute(0, 2, 1))
        ogpu = torch_layers(ngpu, x.permute(0, 2, 1))
        ocpu = torch_layers(ncpu, x.cpu().permute(0, 2, 1))
        pointnet.Z1_FREE = False  # (this tool looks at every layer's output)
        y, saved = pointnet.forward_impl(net, x.contiguous(), True)
    hip = [z.view(B, N, -1).permute(0, 2, 1) for z in saved["zc"]] + list(saved["zf"]) + [y]
    names = ["conv1", "conv2", "conv3", "conv4", "conv5", "fc1", "fc2", "fc3", "fc4"]
    print("%-6s %12s %12s %12s" % ("layer", "hip", "torch-gpu", "torch-cpu"))
    for n, h, a, c, d in zip(names, hip, ogpu, ocpu, o64):
        print("%-6s %12.3e %12.3e %12.3e" % (n, rel(h, d), rel(a, d), rel(c, d)))


if __name__ == "__main__":
    main()
