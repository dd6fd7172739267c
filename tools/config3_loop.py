"""The sampler step of BASELINE configs[3] (reconstruction SampleNet, B = 50 x 2048 points) in a loop, for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o k -- python tools/config3_loop.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from samplenet_amd import SampleNet  # noqa: E402
from samplenet_amd.engine import SamplerTrainStep  # noqa: E402
from samplenet_amd.parallel import FlatGradAllReducer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
B, N, M, K = 50, 2048, 64, 8
torch.manual_seed(0)
net = SampleNet(M, 128, group_size=K, input_shape="bnc", output_shape="bnc", conv_widths=(64, 128, 128, 256), fc_widths=(256, 256),
                fc_batchnorm=False, temperature_floor=1e-2, min_sigma=0.0).to(dev).train()
x = torch.rand(B, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(13)) - 0.5
st = SamplerTrainStep(net, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=FlatGradAllReducer(net), use_graph=False)
for _ in range(steps):
    loss = st(x)
torch.cuda.synchronize()
print("loss", float(loss))
