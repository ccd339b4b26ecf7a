import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
from dbg_cert import cert
g = torch.Generator().manual_seed(3)
for (B, T, C, L, sc) in [(2, 16, 8, 3, 1.0), (2, 16, 8, 3, 0.0), (2, 32, 8, 3, 1.0), (2, 8, 8, 3, 1.0), (2, 16, 8, 0, 1.0)]:
    x = torch.randn(B, T, C, generator=g).cuda() * sc
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    tg = E.targets_on_device(targets, x.device)
    scale, _, coef = E.loss_factors(tg, "mean")
    dx = torch.empty_like(x)
    ws2, nll2, loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
    torch.cuda.synchronize()
    z2, zmm = cert(ws2, B, T, tg.max_len)
    print(f"T={T} C={C} L={L} scale={sc}: repaired {E.ctc_pipeline_repaired(ws2, B, T, tg.max_len)} z2 {z2} zmin-z2 {zmm[:,0]-z2} zmax-z2 {zmm[:,1]-z2}")
