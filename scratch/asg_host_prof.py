import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.criterions import asg
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
W = torch.randn(C + 1, C, generator=g).cuda().requires_grad_(True)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
def step():
    x.grad = None; W.grad = None
    asg.ASGLoss(x, W, targets).backward()
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host per step %.1f us, with sync %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
