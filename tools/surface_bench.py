"""Driver-independent timing of the module-surface legs alone (bench.time_module_surface): python tools/surface_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

out = bench.time_module_surface(torch.device("cuda:0"), 32, 1024, 64, 8, steps=int(os.environ.get("STEPS", "200")))
print(json.dumps({k: v for k, v in out.items() if k != "task_roofline"}, indent=1))
