"""Host cost of a criterion step: the step at a size where the GPU has next to nothing to do (wall time per step = host time)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.criterions import asg, ctc, transducer as TR
def timeit(step, n=300):
    for _ in range(30): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
g = torch.Generator().manual_seed(0)
for (B, T, C, L) in [(8, 40, 100, 5), (128, 40, 100, 5)]:
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    W = torch.randn(C + 1, C, generator=g).cuda().requires_grad_(True)
    tg = torch.randint(C - 2, (B, L), generator=g).tolist()
    def s_asg():
        x.grad = None; W.grad = None
        asg.ASGLoss(x, W, tg).backward()
    def s_ctc():
        x.grad = None
        ctc.CTCLoss(x, tg, C - 1).backward()
    xn = x.detach()
    def s_asg_engine():
        x.grad = None; W.grad = None
        asg.ASGLoss(x * 1.0, W * 1.0, tg).backward()
    print(f"B={B} T={T}: asg {timeit(s_asg):.1f} us  asg through engine {timeit(s_asg_engine):.1f} us  ctc {timeit(s_ctc):.1f} us")
toks = [(i,) for i in range(C)]
crit = TR.Transducer(toks, {i: i for i in range(C)}, blank="optional", allow_repeats=False, reduction="mean")
x = torch.randn(8, 40, C + 1, generator=g).cuda().requires_grad_(True)
tgt = [torch.tensor(t) for t in torch.randint(C, (8, 5), generator=g).tolist()]
def s_tr():
    x.grad = None
    crit(x, tgt).backward()
print(f"transducer B=8: {timeit(s_tr):.1f} us")
