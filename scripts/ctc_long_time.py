import sys, time
sys.path.insert(0, "/root/repo")
import torch
from gtn_applications_amd.criterions import ctc
T, C, B = 1000, 100, 128
for L in (63, 64, 100, 200):
    g = torch.Generator().manual_seed(4)
    x = torch.log_softmax(torch.randn(B, T, C, generator=g), 2).cuda().requires_grad_(True)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    def step():
        x.grad = None
        ctc.CTCLoss(x, targets, C - 1, "mean").backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    print(f"B={B} L={L}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per step")
