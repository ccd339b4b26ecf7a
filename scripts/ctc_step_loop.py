"""scratch (GPU box): N back-to-back cfg2 CTC steps through the C ABI, nothing else -- for rocprofv3 kernel traces of
two builds of the library on one box (WFL_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gtn_applications_amd import engine as E
B, T, C, L = int(os.environ.get("B", 128)), 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda()
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "mean")
dx = torch.empty_like(x)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True, shared_ws=True)
torch.cuda.synchronize()
