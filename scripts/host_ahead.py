"""Is the host ahead of the GPU at the benchmark size?  Host time of the first k steps after a synchronize (empty queues:
no back-pressure) against the GPU-paced time per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.criterions import asg
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
W = torch.randn(C + 1, C, generator=g).cuda().requires_grad_(True)
tg = torch.randint(C - 2, (B, L), generator=g).tolist()
def step():
    x.grad = None; W.grad = None
    asg.ASGLoss(x, W, tg).backward()
for _ in range(20): step()
for k in (1, 2, 4, 8, 16, 64):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"k={k:3d}: host {1e6*(t1-t0)/k:7.1f} us/step   until the GPU is done {1e6*(t2-t0)/k:7.1f} us/step")
# per-call host times inside one step (queues empty)
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(35)
