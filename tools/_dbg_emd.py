import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samplenet_amd._lib import lib, ptr, check
from samplenet_amd import ops
for (b, n, m) in [(2, 256, 64), (1, 128, 128), (2, 2048, 2048)]:
    g = torch.Generator(device="cuda").manual_seed(n + m)
    x1 = torch.rand(b, n, 3, device="cuda", generator=g)
    x2 = torch.rand(b, m, 3, device="cuda", generator=g)
    for name, fn in (("exact", lib.sn_emd_loss), ("fast", lib.sn_emd_loss_fast)):
        ws = torch.zeros(max(1, lib.sn_workspace_bytes(b"emd_loss", b, n, m, 0) // 4), device="cuda")
        cost = torch.empty(b, device="cuda"); g1 = torch.empty_like(x1); g2 = torch.empty_like(x2)
        check(fn(b, n, m, ptr(x1), ptr(x2), ptr(cost), ptr(g1), ptr(g2), ptr(ws), None), name)
        torch.cuda.synchronize()
        per = (n + m) * 11
        w = ws[:per].cpu()
        remL, remR = w[:n], w[n:n + m]
        rL = w[n + m:n + m + 10 * n].view(10, n); rR = w[n + m + 10 * n:per].view(10, m)
        print(b, n, m, name, "cost", cost.tolist()[:2], "nan: remL %d remR %d" % (remL.isnan().sum(), remR.isnan().sum()),
              "rL per level", rL.isnan().sum(1).tolist(), "rR", rR.isnan().sum(1).tolist(), "g1 nan", int(g1.isnan().sum()), "g2", int(g2.isnan().sum()))
        print("   rL[0,:4]", rL[0, :4].tolist(), "rR[0,:4]", rR[0, :4].tolist(), "rL[9,:4]", rL[9,:4].tolist())
