import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gtn_applications_amd._native as N
import numpy as np, torch
from gtn_applications_amd import engine as E
from dbg_cert import cert
B, T, C, L = 1, 8, 8, 1
x = torch.zeros(B, T, C).cuda()
targets = [[2]]
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "mean")
dx = torch.empty_like(x)
ws2, nll2, loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
torch.cuda.synchronize()
z2, zmm = cert(ws2, B, T, tg.max_len)
print("z2", z2, "zmm", zmm, "nll", nll2)
