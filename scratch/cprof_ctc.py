"""scratch: where the host time of CTCLoss(x, targets, blank).backward() goes (same targets: caches hit)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gtn_applications_amd.criterions import ctc
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
def step():
    x.grad = None
    loss = ctc.CTCLoss(x, targets, C - 1, "none")
    loss.backward()
for _ in range(50): step()
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host per step %.1f us, with sync %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# forward only / backward only split
t0 = time.perf_counter()
for _ in range(N):
    loss = ctc.CTCLoss(x, targets, C - 1, "none")
t1 = time.perf_counter(); torch.cuda.synchronize()
print("forward only host %.1f us" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
