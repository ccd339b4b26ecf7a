import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.criterions import asg
SHAPES = [(128, 1000, 100), (128, 1000, 129), (128, 1000, 150), (128, 1000, 190), (128, 1000, 200), (32, 250, 1000)]
if len(sys.argv) > 1:  # B,T,C [B,T,C ...]
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, T, C) in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    W = torch.randn(C + 1, C, generator=g).cuda().requires_grad_(True)
    tg = torch.randint(C - 2, (B, 44), generator=g).tolist()
    m = asg.ASG(C - 1, 1, False).cuda() if False else None
    def step():
        x.grad = None; W.grad = None
        asg.ASGLoss(x, W, tg).backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    from gtn_applications_amd import engine as E
    xd, Wd = x.detach(), W.detach()
    E.dense_viterbi(xd, Wd); torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(3): E.dense_viterbi(xd, Wd)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"B={B} T={T} C={C}: fwd+bwd {(t1-t0)/5*1e3:.2f} ms  viterbi {(t3-t2)/3*1e3:.2f} ms")
