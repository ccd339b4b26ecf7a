import sys, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo")
from gtn_applications_amd.criterions import asg
torch.manual_seed(0)
B, T, C, L = 32, 250, 100, 44
crit = asg.ASG(C - 2, 1, True).cuda()
x = torch.randn(B, T, C).cuda().requires_grad_(True)
batches = [[torch.randint(C - 2, (L,)) for _ in range(B)] for _ in range(300)]
it = iter(batches)
def step():
    x.grad = None; crit.transitions.grad = None
    crit(x, next(it)).backward()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"fresh targets: host {(t1-t0)/100*1e3:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
