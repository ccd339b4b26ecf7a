"""CTC with targets of 100 / 200 labels (the two-to-four-positions-per-lane kernels): posterior error against the float64
oracle and step time at T = 1000, C = 100."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gtn_applications_amd.criterions import ctc
from oracle import recurrences as OR
T, C = 1000, 100
for (B, L, s) in ((16, 100, 1.0), (16, 100, 2.5), (16, 200, 1.0)):
    g = torch.Generator().manual_seed(3)
    lp = torch.log_softmax(s * torch.randn(B, T, C, generator=g), 2)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    want_loss, dlp = OR.ctc_loss_grad_batched(lp.numpy(), targets, C - 1)
    lpg = lp.cuda().requires_grad_(True)
    l2 = ctc.CTCLoss(lpg, targets, C - 1, "none").sum(); l2.backward()
    e = (lpg.grad.cpu().double() - torch.tensor(dlp)).abs() * B
    print(f"L={L} spread {s}: max posterior err {e.max():.2e}, median over utterances {e.amax(dim=(1,2)).median():.2e}")
B, L = 128, 100
g = torch.Generator().manual_seed(4)
x = torch.log_softmax(torch.randn(B, T, C, generator=g), 2).cuda().requires_grad_(True)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
def step():
    x.grad = None
    ctc.CTCLoss(x, targets, C - 1, "mean").backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print(f"B={B} L={L}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step")
