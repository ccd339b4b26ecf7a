import sys, os, time, torch
sys.path.insert(0, "/root/repo")
os.environ.setdefault("WFL_DENSE_NGRAM", "0")   # the general lattice path for the n-gram models (epsilon arcs)
from gtn_applications_amd.criterions import transducer as TR
B, n = 16, 2
N, T, L = 81, 250, 44
torch.manual_seed(0)
crit = TR.Transducer([(i,) for i in range(N)], {i: i for i in range(N)}, ngram=n, reduction="mean", blank="optional", allow_repeats=False).cuda()
x = torch.randn(B, T, N + 1).cuda().requires_grad_(True)
targets = [t.squeeze() for t in torch.randint(N, size=(B, L)).split(1)]
def step():
    x.grad = None; crit.transition_params.grad = None
    crit(x, targets).backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(f"general lattice path, bigram, B={B}: {(time.perf_counter()-t0)/20*1e3:.3f} ms per step")
