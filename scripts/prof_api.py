import sys, cProfile, pstats, torch
sys.path.insert(0, "/root/repo")
from gtn_applications_amd.criterions import ctc
g = torch.Generator().manual_seed(0)
B, T, C, L = 128, 1000, 100, 44
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
def step():
    x.grad = None
    ctc.CTCLoss(x, targets, C - 1).backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
