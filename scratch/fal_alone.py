"""cfg3's force-aligned numerator alone (no denominator sweeps beside it): per-kernel times via rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gtn_applications_amd import engine as E
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda()
W = (0.1 * torch.randn(C + 1, C, generator=g)).cuda()
tg = E.targets_on_device(torch.randint(C - 2, (B, L), generator=g).tolist(), x.device)
pack = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, x.device)
scale, cpos, cneg = E.loss_factors(tg, "mean")
dx = torch.empty_like(x); dW = torch.zeros_like(W)
for i in range(30):
    fal = E.lattice_forward(x, pack, weights=W, need_beta=True)
    E.lattice_grad(fal, cneg, coef_w=cneg, gout=None, dx=dx, accumulate=False, dW=dW)
torch.cuda.synchronize()
