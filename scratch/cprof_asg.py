import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import asg as AS, ctc
B, T, C, L, N = 128, 1000, 100, 44, 300
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
Wt = torch.zeros(C + 1, C, device="cuda", requires_grad=True)
batches = [torch.randint(C - 2, (B, L), generator=g).tolist() for _ in range(N + 20)]
SAME = len(sys.argv) > 2
def afwd(i):
    i = 0 if SAME else i
    x.grad = None; Wt.grad = None
    AS.ASGLoss(x, Wt, batches[i], "mean").backward()
def cfwd(i):
    x.grad = None
    ctc.CTCLoss(x, batches[i], C - 1).backward()
which = sys.argv[1] if len(sys.argv) > 1 else "asg"
fn = afwd if which == "asg" else cfwd
for i in range(20): fn(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20, 20 + N): fn(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
