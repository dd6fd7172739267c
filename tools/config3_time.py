"""Wall-clock of bench.py's config3_sampler leg alone (eager / graph / script), one JSON line:  python tools/config3_time.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

out = bench.time_config3_sampler(torch.device("cuda:0"))
print(json.dumps({k: (round(v["ms_per_step"], 4) if isinstance(v, dict) else v) for k, v in out.items() if k != "workload"}))
