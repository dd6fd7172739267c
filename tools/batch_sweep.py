"""python tools/batch_sweep.py [B ...]: bench.time_batch_sweep alone (whole sampler step at growing batches)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

bs = tuple(int(a) for a in sys.argv[1:]) or (32, 128, 512, 2048)
if os.environ.get("PERSIST_MIN") is not None:  # A/B of the persistent forward GEMMs (0: off)
    from samplenet_amd._lib import lib

    lib.sn_conv_stack_set_persist_min_tiles(int(os.environ["PERSIST_MIN"]))
for row in bench.time_batch_sweep(torch.device("cuda:0"), 1024, 64, 8, batches=bs):
    print(json.dumps(row))
