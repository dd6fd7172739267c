#!/usr/bin/env python3
"""python tools/variant_loop.py config3_sampler|config5_progressive [steps]: the secondary bench legs' captured steps in a loop --
for rocprofv3 --kernel-trace --stats (profiles/rNN/<leg>_kernel_stats.csv)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

leg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fn = {"config3_sampler": bench.time_config3_sampler, "config5_progressive": bench.time_config5_progressive}[leg]
out = fn(torch.device("cuda:0"), steps=steps)
print({k: v for k, v in out.items() if isinstance(v, dict)})
