import torch, numpy as np, sys
sys.path.insert(0, "/root/repo")
from gtn_applications_amd.criterions import ctc
from oracle import recurrences as OR
g = torch.Generator().manual_seed(0)
B, T, C, L = 128, 1000, 100, 44
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
tgt = torch.randint(C - 2, (B, L), generator=g)
loss = ctc.CTCLoss(x, tgt.tolist(), C - 1)
loss.backward()
rows = x.grad.sum(dim=2) * (-B)
print("row sums: min %.6f max %.6f mean %.6f" % (rows.min().item(), rows.max().item(), rows.mean().item()))
bad = (rows - 1).abs()
print("max dev", bad.max().item(), "at", np.unravel_index(bad.argmax().item(), rows.shape))
xs = x.detach()[:2].cpu().double().numpy()
wl, wdx = OR.ctc_loss_grad(xs, tgt[:2].tolist(), C - 1)
got = x.grad[:2].cpu().double().numpy() * (B / 2.0)
err = np.abs(got - wdx)
print("grad vs oracle: max abs err %.3e, max |grad| %.3e, max rel err where |g|>1e-4: %.3e" % (err.max(), np.abs(wdx).max(), (err / np.maximum(np.abs(wdx), 1e-30))[np.abs(wdx) > 1e-4].max()))
