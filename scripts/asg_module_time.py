import sys, time, torch
sys.path.insert(0, "/root/repo")
from gtn_applications_amd.criterions import asg
torch.manual_seed(0)
for B in (32, 128):
    T, C, L = 1000 if B == 128 else 250, 100, 44
    crit = asg.ASG(C - 2, 1, True).cuda()
    x = torch.randn(B, T, C).cuda().requires_grad_(True)
    targets = [torch.randint(C - 2, (L,)) for _ in range(B)]
    def step():
        x.grad = None; crit.transitions.grad = None
        crit(x, targets).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"ASG module fwd+bwd B={B} T={T}: host {(t1-t0)/50*1e3:.3f} ms, with the drain {(t2-t0)/50*1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(200): asg.pack_targets_batch(targets, 1, C - 1)
    print(f"   target preparation (batched): {(time.perf_counter()-t0)/200*1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(50):
        tg = [asg.pack_replabels(t.tolist(), 1) for t in targets]
        for i, g in enumerate(tg):
            inter = [C - 1] * (2 * len(g) + 1); inter[1::2] = g; tg[i] = inter
    print(f"   target preparation (row by row): {(time.perf_counter()-t0)/50*1e3:.3f} ms")
