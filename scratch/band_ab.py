"""scratch: FAL numerator gradient, banded kernel vs general kernel (run twice: WFL_LATTICE_BAND_GRAD unset / =0), saved to npz."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtn_applications_amd import engine as E
out = sys.argv[1]
B, T, C, L = 16, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g)
W = torch.randn(C + 1, C, generator=g)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
targets[3] = targets[3][:7]
targets[5] = [4, 4, 4, 9, 9, 4]
dev = torch.device("cuda")
xg, Wg = x.to(dev), W.to(dev)
coef = torch.rand(B, generator=g).to(dev) + 0.5
tg = E.targets_on_device(targets, dev)
pack = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, dev)
fal = E.lattice_forward(xg, pack, weights=Wg)
dx = torch.zeros_like(xg)
dW = torch.zeros_like(Wg)
E.lattice_grad(fal, coef, coef_w=coef, gout=None, dx=dx, dW=dW)
torch.cuda.synchronize()
np.savez(out, dx=dx.cpu().numpy(), dW=dW.cpu().numpy(), logz=fal.logz.cpu().numpy())
