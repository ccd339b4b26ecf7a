import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gtn_applications_amd import engine as E
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
for kind in ("randn", "log_softmax(randn)"):
    for s in (1.0, 1.5, 2.0, 3.0, 5.0):
        x = s * torch.randn(B, T, C, generator=g)
        if kind != "randn":
            x = torch.log_softmax(x, 2)
        xd = x.cuda()
        tg = E.targets_on_device(targets, xd.device)
        scale, _, coef = E.loss_factors(tg, "none")
        dx = torch.empty_like(xd)
        ws, nll = E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx)
        torch.cuda.synchronize()
        print(kind, "scale", s, "repaired", E.ctc_pipeline_repaired(ws, B, T, tg.max_len), "of", B)
