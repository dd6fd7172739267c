#!/usr/bin/env python3
"""python tools/cls_loop.py [steps] [K]: the classification sampler's captured training step (bench.py's config1_classification `graph`
leg alone) in a loop -- for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from samplenet_amd import SampleNet  # noqa: E402
from samplenet_amd.engine import SamplerTrainStep  # noqa: E402
from samplenet_amd.parallel import FlatGradAllReducer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = SampleNet(64, 128, group_size=K, input_shape="bnc", output_shape="bnc", last_fc_batchnorm=True, min_sigma=0.0).to(dev).train()
x = torch.rand(32, 1024, 3, device=dev) - 0.5
st = SamplerTrainStep(net, x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, reducer=FlatGradAllReducer(net), use_graph=True, input_ring=[x])
for _ in range(steps):
    loss = st.replay(0)
torch.cuda.synchronize()
print("loss", float(loss), "fast", st._fast_path())
